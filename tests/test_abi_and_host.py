"""CPU-only checks of the boundary and the host logic (no GPU compute calls)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from m6anet_amd import _lib, constants, engine, synthetic

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    h = open(os.path.join(REPO, "include", "m6a.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(m6a_[a-z_0-9]+)\s*\(", h)))


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    declared = header_symbols()
    assert declared, "no declarations parsed from include/m6a.h"
    for name in declared:
        assert hasattr(L, name), "libm6a_hip.so does not export %s" % name
    assert sorted(_lib.SYMBOLS) == declared            # the binding knows exactly the header's surface
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r" T (m6a_[a-z_0-9]+)", out)))
    assert exported == declared, (exported, declared)


def test_version_and_errors_without_gpu():
    L = _lib.load()
    assert b"gfx950" in L.m6a_version()
    h = C.c_void_p()
    w = np.zeros(7997, np.float32)
    assert L.m6a_create(C.byref(h), w.ctypes.data, 10, 0) == -1          # M6A_EINVAL: wrong blob size
    assert b"7997" in L.m6a_last_error(None)
    big = w.copy()
    big[3132 + 77] = 1e19                                                # a layer-2 weight the 2^64 scaling would overflow
    assert L.m6a_create(C.byref(h), big.ctypes.data, 7997, 0) == -6 and b"2^63" in L.m6a_last_error(None)   # M6A_EUNSUPPORTED
    import torch
    if not torch.cuda.is_available():
        rc = L.m6a_create(C.byref(h), w.ctypes.data, 7997, 0)
        assert rc == -5 and not h.value                                  # M6A_ENODEV, loud, no fallback
        with pytest.raises(_lib.M6AError):
            engine.M6ANetEngine(weights=w)


def test_vocabulary_matches_reference_capture():
    want = open(os.path.join(REPO, "tests", "golden", "vocab66.txt")).read().split()
    assert constants.ALL_KMERS == want
    assert len(constants.ALL_7MERS) == 288 and len(constants.M6A_KMERS) == 18
    assert constants.kmer7_to_ids("AGGACTT") == [constants.KMER_TO_INT[k] for k in ("AGGAC", "GGACT", "GACTT")]


def test_flush_groups_helper_matches_reference_loop():
    def ref(S, bs, spb):      # the loop shape of m6anet/utils/inference_utils.py:33,47
        nb, out, start = (S + bs - 1) // bs, [0], 0
        for it in range(nb):
            if (it + 1) % spb:
                out.append(min((it + 1) * bs, S))
                start = it + 1
        if start < nb:
            out.append(S)
        return out
    for S, bs, spb in [(101, 16, 2), (101, 51, 2), (101, 13, 2), (101, 8, 3), (40, 16, 1), (5, 16, 2), (1, 1, 2),
                       (1000, 7, 5), (64, 16, 2), (0, 16, 2)]:
        assert engine.flush_groups(S, bs, spb).tolist() == ref(S, bs, spb), (S, bs, spb)


@pytest.mark.parametrize("n_shards", [1, 2, 3, 8])
def test_shard_plan_is_group_aligned_and_balanced(n_shards):
    g = np.random.Generator(np.random.PCG64(n_shards))
    for bags in (np.full(5000, 20), g.integers(20, 400, size=3000)):
        off = np.concatenate([[0], np.cumsum(bags)]).astype(np.int64)
        cuts = engine.shard_plan(off, n_shards)
        groups = set(engine.flush_groups(len(bags)).tolist())
        assert cuts[0] == 0 and cuts[-1] == len(bags) and np.all(np.diff(cuts) >= 0)
        assert all(int(c) in groups for c in cuts)
        reads = np.diff(off[cuts])
        assert reads.max() - reads.min() <= 2 * 32 * bags.max()       # within a couple of groups of even


def test_weights_blobs_and_state_dict_order():
    for name, (blob, thr, norm) in constants.PRETRAINED_CONFIGS.items():
        w = engine.load_weights(name)
        assert w.shape == (7997,) and np.all(np.isfinite(w))
        assert os.path.exists(constants.asset_path(norm))
    sd = {k: np.arange(n, dtype=np.float32) for k, n in [
        ("read_level_encoder.1.embedding_layer.weight", 132), ("read_level_encoder.3.layers.0.weight", 2250),
        ("read_level_encoder.3.layers.0.bias", 150), ("read_level_encoder.3.layers.1.weight", 150),
        ("read_level_encoder.3.layers.1.bias", 150), ("read_level_encoder.3.layers.1.running_mean", 150),
        ("read_level_encoder.3.layers.1.running_var", 150), ("read_level_encoder.4.layers.0.weight", 4800),
        ("read_level_encoder.4.layers.0.bias", 32), ("pooling_filter.probability_layer.0.weight", 32),
        ("pooling_filter.probability_layer.0.bias", 1)]}
    w = engine.weights_from_state_dict(sd)
    assert w.size == 7997 and w[132] == 0 and w[131] == 131 and w[-1] == 0
    with pytest.raises(ValueError):
        engine.load_weights("nope")


def test_synthetic_generator_is_deterministic():
    a = synthetic.make_sites(300, (50, 500), seed=7)
    b = synthetic.make_sites(300, (50, 500), seed=7)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert a["X"].dtype == np.float32 and np.abs(a["X"]).max() <= 6.0
    assert a["site_kmers"].max() < 66 and a["off"][0] == 0 and np.diff(a["off"]).min() >= 50


def test_reference_written_sites_matches_the_reference_runs(golden):
    """m6a_reference_written_sites against the rows six real `m6anet inference` runs wrote (the `_written` masks
    captured by tests/golden/make_golden.py): the reference writes a prefix of the sites -- everything up to its
    last flush (m6anet/utils/inference_utils.py:47)."""
    import re
    g = golden("bundled_site.npz")
    keys = [k for k in g.files if k.endswith("_written")]
    assert len(keys) >= 6
    for k in keys:
        bs, spb = (int(x) for x in re.match(r"T\d+_bs(\d+)_spb(\d+)_", k).groups())
        mask = g[k].astype(bool)
        n = engine.reference_written_sites(mask.size, bs, spb)
        assert np.array_equal(mask, np.arange(mask.size) < n), (k, n, int(mask.sum()))
    assert engine.reference_written_sites(101, 51, 2) == 51          # batch 51 -> 51 of 101
    assert engine.reference_written_sites(101, 26, 2) == 78
    assert engine.reference_written_sites(101, 16, 2) == 101
    assert engine.reference_written_sites(0, 16, 2) == 0
    assert engine.reference_written_sites(32, 16, 1) == 0            # save_per_batch 1: the reference never flushes


def test_multi_gpu_launcher_stops_the_job_when_a_rank_dies():
    """`inference --gpus N`: the launcher waits for its ranks; the first one that fails ends the job -- the others
    (which would wait for it in the exchange) are terminated by pid -- and its exit code is the job's."""
    import subprocess
    import sys
    import time
    from m6anet_amd import multi_gpu
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, "-c", "import time; time.sleep(120)"]),
             subprocess.Popen([sys.executable, "-c", "import sys, time; time.sleep(0.3); sys.exit(7)"])]
    assert multi_gpu._wait_all(procs) == 7
    assert time.time() - t0 < 30 and all(p.poll() is not None for p in procs)
    procs = [subprocess.Popen([sys.executable, "-c", "pass"]) for _ in range(3)]
    assert multi_gpu._wait_all(procs) == 0


def test_multi_gpu_rank_command_line_round_trips():
    """A rank is started with the launcher's own options: parsing the rebuilt command line gives the same namespace."""
    from argparse import ArgumentParser
    from m6anet_amd import multi_gpu
    from m6anet_amd.scripts import inference
    p = ArgumentParser(parents=[inference.argparser()])
    argv = ["--input_dir", "a", "b", "--out_dir", "o", "--gpus", "4", "--num_iterations", "77", "--seed", "3", "--batch_size", "8",
            "--drop_unflushed_tail", "--read_proba_threshold", "0.0125", "--model_config", "m.toml"]
    a = p.parse_args(argv)
    assert vars(p.parse_args(multi_gpu.rank_argv(a))) == vars(a)
    a = p.parse_args(["--input_dir", "x.m6astore", "--out_dir", "o"])
    assert vars(p.parse_args(multi_gpu.rank_argv(a))) == vars(a)


def test_mt19937_jump_polynomials_against_numpy():
    """assets/mt19937_jump.bin (tools/make_mt_jump.py; embedded in libm6a_hip.so for the segmented stream generator):
    r = t^(i*G - 512) mod phi means  out[k + i*G - 512] = XOR over the set coefficients t of r of out[k + t]  for NumPy's
    legacy stream (tempering is linear, so the relation holds for the outputs).  Checked here for the first polynomials of
    the 2^16 and 2^20 regimes and the header for all three."""
    import struct
    from m6anet_amd.constants import asset_path
    blob = open(asset_path("mt19937_jump.bin"), "rb").read()
    assert blob[:8] == b"M6AMTJP1"
    n_reg, n_per, words, back = struct.unpack("<IIII", blob[8:24])
    assert (n_reg, n_per, words, back) == (3, 31, 312, 512) and len(blob) == 24 + n_reg * (8 + n_per * words * 8)
    per = 8 + n_per * words * 8
    out = np.frombuffer(np.random.RandomState(2024).bytes(4 * (3 * (1 << 20) + 22000)), dtype=np.uint32)
    Gs = []
    for r in range(n_reg):
        base = 24 + r * per
        G, = struct.unpack("<Q", blob[base:base + 8])
        Gs.append(G)
        for i in (1, 2, 3) if r < 2 else ():
            poly = np.frombuffer(blob, dtype=np.uint64, count=words, offset=base + 8 + (i - 1) * words * 8)
            bits = np.unpackbits(poly.view(np.uint8), bitorder="little")
            taps = np.flatnonzero(bits)
            assert taps.max() < 19937 and 1000 < taps.size < 11000
            D = i * G - back
            if D + 400 + 19937 > out.size:
                continue
            ks = np.arange(0, 300)
            acc = np.zeros(ks.size, np.uint32)
            for t in taps:
                acc ^= out[ks + t]
            assert np.array_equal(acc, out[ks + D]), (G, i)
    assert Gs == [1 << 16, 1 << 20, 1 << 24]


def test_multi_gpu_exchange_plumbing(tmp_path, monkeypatch):
    """The pieces of `inference --gpus N` that need no GPU: more ranks than visible devices is refused (there are none
    here), an unknown transport is an error, a rank never waits for ever on a peer, results are published atomically."""
    import os
    from m6anet_amd import multi_gpu
    monkeypatch.delenv("M6A_EXCHANGE", raising=False)
    monkeypatch.delenv("M6A_SHARE_GPU", raising=False)
    with pytest.raises(RuntimeError, match="HIP device"):
        multi_gpu.exchange_mode(2)
    monkeypatch.setenv("M6A_EXCHANGE", "carrier-pigeon")
    with pytest.raises(ValueError):
        multi_gpu.exchange_mode(2)
    monkeypatch.setenv("M6A_EXCHANGE", "host")
    assert multi_gpu.exchange_mode(64) == ("host", True)
    # the default: NO device exchange (every rank writes its own rows); sharing a GPU is its own switch, and RCCL refuses it
    monkeypatch.delenv("M6A_EXCHANGE")
    monkeypatch.setenv("M6A_SHARE_GPU", "1")
    assert multi_gpu.exchange_mode(8) == ("none", True)
    monkeypatch.setenv("M6A_EXCHANGE", "rccl")
    with pytest.raises(ValueError, match="one GPU per rank"):
        multi_gpu.exchange_mode(2)
    monkeypatch.delenv("M6A_SHARE_GPU")
    monkeypatch.setenv("M6A_EXCHANGE_TIMEOUT", "0.2")
    with pytest.raises(TimeoutError, match="rank 3"):
        multi_gpu._wait_for(str(tmp_path / "never"), "rank 3's results", os.getppid())
    with pytest.raises(RuntimeError, match="launcher has gone away"):
        multi_gpu._wait_for(str(tmp_path / "never"), "anything", os.getppid() + 12345)
    p = str(tmp_path / "shard1.bin")
    multi_gpu._publish(p, b"abc" * 1000)
    assert open(p, "rb").read() == b"abc" * 1000 and os.listdir(tmp_path) == ["shard1.bin"]
    multi_gpu._wait_for(p, "present", os.getppid())


def test_bundled_weights_scale_exactly(weights):
    """The encoder's scalings by powers of two (m6a_api.hip::build_fragments) must be exact for the kernels to form the
    reference's products bit for bit: layer 1's batch norm + ReLU is a clamped fma on (alpha, beta) * 2^-64 with W2 and its
    bias column * 2^64; layer 2's ReLU is x + |x| = 2 relu(x) with W3 * 0.5.  Exact iff nothing overflows or turns
    sub-normal: true for the four bundled checkpoints (alpha, beta of a trained batch norm are nowhere near 2^-62)."""
    f32 = np.float32
    for name, w in weights.items():
        g, be, mu, var = w[2532:2682], w[2682:2832], w[2832:2982], w[2982:3132]       # blob layout: include/m6a.h
        alpha = (g * (f32(1) / np.sqrt(var + f32(1e-5))).astype(f32)).astype(f32)
        beta = (be.astype(np.float64) - mu.astype(np.float64) * alpha.astype(np.float64)).astype(f32)   # fma(-mean, alpha, bias)
        for v in (alpha, beta):
            s = (v * f32(2.0 ** -64)).astype(f32)
            assert np.array_equal((s * f32(2.0 ** 64)).astype(f32), v), name
            assert np.abs(v[v != 0]).min() > 2.0 ** -40, (name, float(np.abs(v[v != 0]).min()))
        w2b2 = w[3132:7964]                                                               # W2 [32,150], b2 [32]
        up = (w2b2 * f32(2.0 ** 64)).astype(f32)
        assert np.isfinite(up).all() and np.array_equal((up * f32(2.0 ** -64)).astype(f32), w2b2), name
        w3 = w[7964:7996]
        assert np.array_equal((w3 * f32(0.5)).astype(f32) * f32(2.0), w3), name


def test_early_rank_start_helpers(tmp_path, monkeypatch):
    """m6anet_amd/_early.py (standard library only: it runs before the heavy imports of `python -m m6anet_amd`): option
    scanning, the exchange directory's home, and the cases in which it must NOT start anything."""
    from m6anet_amd import _early
    argv = ["--input_dir", "a", "b", "--out_dir", "o", "--gpus=3", "--num_iterations", "5"]
    assert _early._values(argv, "--input_dir") == ["a", "b"] and _early._values(argv, "--gpus") == ["3"]
    assert _early._values(argv, "--out_dir") == ["o"] and _early._values(argv, "--seed") is None
    (tmp_path / "d").mkdir()
    (tmp_path / "d" / "data.json").write_bytes(b"x" * 1000)
    assert _early.store_size_estimate([tmp_path / "d"]) == 600 + (64 << 20)
    # a store that fits nowhere but the output directory's file system ends there or nowhere; an override wins
    assert _early.exchange_base(1 << 62, str(tmp_path)) is None
    assert _early.exchange_base(0, str(tmp_path)) in ("/dev/shm", __import__("tempfile").gettempdir(), str(tmp_path))
    monkeypatch.setenv("M6A_XDIR_BASE", str(tmp_path))
    assert _early.exchange_base(1 << 62, "/nonexistent") == str(tmp_path)
    monkeypatch.delenv("M6A_XDIR_BASE")
    # nothing is started: not inference, one GPU, a rank itself, help, more ranks than render nodes without the host transport
    for av, env in ((["dataprep", "--gpus", "2"], {}), (["inference", "--input_dir", "a", "--out_dir", "o"], {}),
                    (["inference", "--input_dir", "a", "--out_dir", "o", "--gpus", "1"], {}),
                    (["inference", "--input_dir", "a", "--out_dir", "o", "--gpus", "2"], {"M6A_RANK": "1"}),
                    (["inference", "--input_dir", "a", "--out_dir", "o", "--gpus", "2", "--help"], {}),
                    (["inference", "--input_dir", "a", "--out_dir", "o", "--gpus", "63"], {}),
                    (["inference", "--input_dir", "a", "--gpus", "2"], {"M6A_EXCHANGE": "host"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        _early.maybe_start(av)
        assert _early.state is None, av
        for k in env:
            monkeypatch.delenv(k)
    # the early device estimate honours the *_VISIBLE_DEVICES variables (never more than the list names)
    n_all = _early.visible_devices_estimate()
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    assert _early.visible_devices_estimate() == min(n_all, 1)
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "")
    assert _early.visible_devices_estimate() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    assert _early.strip_command(["inference", "--gpus", "2"]) == ["--gpus", "2"] and _early.strip_command(["--gpus", "2"]) == ["--gpus", "2"]
    assert not _early.ranks_may_share_a_gpu()


def test_cpu_baseline_reports_what_the_process_may_use():
    """bench.py's cpu_baseline states its CPUs honestly: effective_cores = affinity mask cut by the cgroup quota (the GPU boxes
    lease 16 of 256 logical CPUs), never more than the host has; the product's own thread counts use the same figure."""
    import bench
    from m6anet_amd import _io
    f = bench.host_cpu_facts()
    assert 1 <= f["effective_cores"] <= f["affinity_cpus"] <= f["logical_cpus"]
    assert f["cgroup_cpu_quota"] is None or f["effective_cores"] <= int(f["cgroup_cpu_quota"] + 0.5) or f["effective_cores"] == 1
    assert _io.usable_cpus() <= f["affinity_cpus"] and abs(_io.usable_cpus() - f["effective_cores"]) <= 1
    cal = bench.cpu_calibration()
    assert cal is None or 0.5 < cal["oracle_over_reference"]["whole_path"] < 2.0
