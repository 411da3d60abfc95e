"""Loader + CSV writers (host side of config #1) on CPU, against what the reference produced."""
import gzip
import os

import numpy as np
import pytest

from m6anet_amd import data_utils, inference_utils

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DATA = os.path.join(GOLD, "ref_tests_data")


def test_loader_reproduces_reference_dataset(golden):
    b = golden("bundled_inputs.npz")
    batch = data_utils.load_sites([DATA], 20, "norm_hct116.npz")
    assert batch.n_sites == 101 and batch.X.shape == (5595, 9) and batch.X.dtype == np.float32
    assert np.array_equal(batch.X, b["X"])                       # same float64 normalise, same float32 cast
    assert np.array_equal(batch.site_kmers, b["site_kmers"])
    assert np.array_equal(batch.off, b["off"])
    assert list(batch.tx_ids) == list(b["tx_ids"]) and np.array_equal(batch.tx_pos, b["tx_pos"])
    assert np.array_equal(np.concatenate(batch.read_ids), b["read_ids"])
    assert batch.kmer5 == [k[1:6] for k in b["kmer7"]]


def test_min_reads_filter():
    batch = data_utils.load_sites(DATA, 100, "norm_hct116.npz")
    assert 0 < batch.n_sites < 101 and batch.n_reads.min() >= 100


def test_csv_rows_are_byte_identical_to_the_reference(golden):
    """Feed the reference's own numbers through our writers: the bytes must match its CSVs."""
    batch = data_utils.load_sites([DATA], 20, "norm_hct116.npz")
    rp = golden("bundled_readprob.npz")["hct116"]
    g = golden("bundled_site.npz")
    site_csv = inference_utils.SITE_HEADER + "".join(
        inference_utils.format_site_rows(batch, g["T5_bs16_spb2_seed0_site"], g["T5_bs16_spb2_seed0_mod"]))
    assert site_csv.encode() == open(os.path.join(GOLD, "config1_site_proba.csv"), "rb").read()
    # per-read file: the reference run encoded per 16-site batch, whose sgemm rounds ~1e-8 away from
    # the one-batch capture in bundled_readprob.npz -- so ids/format byte-exact, numbers to 2e-7
    ours = (inference_utils.INDIV_HEADER + "".join(inference_utils.format_indiv_rows(batch, rp))).splitlines()
    ref = gzip.open(os.path.join(GOLD, "config1_indiv_proba.csv.gz"), "rt").read().splitlines()
    assert len(ours) == len(ref) == 5596 and ours[0] == ref[0]
    for a, b in zip(ours[1:], ref[1:]):
        ia, pa = a.rsplit(",", 1)
        ib, pb = b.rsplit(",", 1)
        assert ia == ib and len(pa) == len(pb) and abs(float(pa) - float(pb)) < 2e-7
    assert ours[1].split(",")[2] == "966210.0"          # read ids print as floats (inference_utils.py:66)


def test_replicate_loader_ids_and_order(tmp_path):
    import shutil
    rep = tmp_path / "rep1"
    rep.mkdir()
    for fn in ("data.info", "data.json"):
        shutil.copyfile(os.path.join(DATA, fn), rep / fn)
    batch = data_utils.load_sites([DATA, str(rep)], 20, "norm_hct116.npz")
    single = data_utils.load_sites([DATA], 1, "norm_hct116.npz")
    # every site now has doubled reads, so sites with >= 10 reads per replicate pass the filter
    assert batch.n_sites == int((single.n_reads * 2 >= 20).sum())
    want = gzip.open(os.path.join(GOLD, "replicate_indiv_proba.csv.gz"), "rt").read().splitlines()[1:]
    ours = [(batch.tx_ids[s], batch.tx_pos[s], rid) for s in range(batch.n_sites) for rid in batch.read_ids[s]]
    ref_ids = [tuple(r.split(",")[:3]) for r in want]
    # the reference run wrote its flushed groups only; ours is a superset in the same order
    assert [(a, str(b), c) for a, b, c in ours[:len(ref_ids)]] == ref_ids
    s0 = batch.read_ids[0]
    assert s0[0].endswith("_0") and s0[-1].endswith("_1")


# ------------------------------------------------------------------ native loader / writers -----
def test_native_symbols_match_header():
    import re
    import subprocess
    from m6anet_amd import _io
    h = open(os.path.join(os.path.dirname(GOLD), "..", "include", "m6a_io.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    declared = sorted(set(re.findall(r"\b(m6a_io_[A-Za-z_0-9]+)\s*\(", h)))
    out = subprocess.run(["nm", "-D", "--defined-only", _io.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r" T (m6a_io_[A-Za-z_0-9]+)", out)))
    assert declared == exported == sorted(_io.SYMBOLS)


def test_native_loader_is_bit_identical_to_python_loader(tmp_path):
    import shutil
    rep = tmp_path / "rep1"
    rep.mkdir()
    for fn in ("data.info", "data.json"):
        shutil.copyfile(os.path.join(DATA, fn), rep / fn)
    for dirs, min_reads in (([DATA], 20), ([DATA], 1), ([DATA, str(rep)], 20)):
        py = data_utils.load_sites(dirs, min_reads, "norm_hct116.npz")
        for threads in (1, 3):
            nat = data_utils.load_sites_native(dirs, min_reads, "norm_hct116.npz", n_threads=threads)
            assert np.array_equal(nat.X, py.X) and nat.X.dtype == np.float32
            assert np.array_equal(nat.site_kmers, py.site_kmers) and np.array_equal(nat.off, py.off)
            assert nat.tx_ids == list(py.tx_ids) and np.array_equal(nat.tx_pos, py.tx_pos) and nat.kmer5 == py.kmer5
            if len(dirs) == 1:
                assert np.array_equal(nat.native.read_id_values, np.concatenate(py.read_ids))
            else:
                ids = ["%d_%d" % (v, r) for v, r in zip(nat.native.read_id_values, nat.native.read_rep)]
                assert ids == [x for site in py.read_ids for x in site]
            nat.native.close()


def test_native_loader_without_normalisation_and_errors(tmp_path):
    from m6anet_amd import _io
    py = data_utils.load_sites([DATA], 20, None)
    nat = data_utils.load_sites_native([DATA], 20, None)
    assert np.array_equal(nat.X, py.X)
    with pytest.raises(_io.M6AIOError):
        data_utils.load_sites_native([str(tmp_path)], 20, None)                 # no data.info
    with pytest.raises(_io.M6AIOError):
        data_utils.load_sites_native([DATA], 100000, None)                      # nothing passes the filter
    bad = tmp_path / "bad"
    bad.mkdir()
    (bad / "data.info").write_text(open(os.path.join(DATA, "data.info")).read())
    (bad / "data.json").write_text(open(os.path.join(DATA, "data.json")).read()[:5000])
    with pytest.raises(_io.M6AIOError):
        data_utils.load_sites_native([str(bad)], 20, None)                      # truncated data.json


def test_native_csv_bytes_equal_python_and_reference(golden, tmp_path):
    import shutil
    g = golden("bundled_site.npz")
    rp = golden("bundled_readprob.npz")["hct116"]
    site, mod = g["T5_bs16_spb2_seed0_site"], g["T5_bs16_spb2_seed0_mod"]
    py = data_utils.load_sites([DATA], 20, "norm_hct116.npz")
    nat = data_utils.load_sites_native([DATA], 20, "norm_hct116.npz")
    out = tmp_path / "o"
    out.mkdir()
    nat.native.write_csv(str(out), rp, site, mod, write_header=True, n_threads=3)
    want_site = inference_utils.SITE_HEADER + "".join(inference_utils.format_site_rows(py, site, mod))
    want_indiv = inference_utils.INDIV_HEADER + "".join(inference_utils.format_indiv_rows(py, rp))
    assert (out / "data.site_proba.csv").read_text() == want_site
    assert (out / "data.indiv_proba.csv").read_text() == want_indiv
    assert (out / "data.site_proba.csv").read_bytes() == open(os.path.join(GOLD, "config1_site_proba.csv"), "rb").read()
    # replicates: "<id>_<rep>" read ids
    rep = tmp_path / "rep1"
    rep.mkdir()
    for fn in ("data.info", "data.json"):
        shutil.copyfile(os.path.join(DATA, fn), rep / fn)
    py2 = data_utils.load_sites([DATA, str(rep)], 20, "norm_hct116.npz")
    nat2 = data_utils.load_sites_native([DATA, str(rep)], 20, "norm_hct116.npz")
    rp2 = np.random.Generator(np.random.PCG64(1)).random(py2.X.shape[0], dtype=np.float32)
    sp2 = np.random.Generator(np.random.PCG64(2)).random(py2.n_sites, dtype=np.float32)
    mr2 = np.random.Generator(np.random.PCG64(3)).random(py2.n_sites)
    nat2.native.write_csv(str(out), rp2, sp2, mr2, write_header=True)
    assert (out / "data.indiv_proba.csv").read_text() == inference_utils.INDIV_HEADER + "".join(inference_utils.format_indiv_rows(py2, rp2))
    assert (out / "data.site_proba.csv").read_text() == inference_utils.SITE_HEADER + "".join(inference_utils.format_site_rows(py2, sp2, mr2))


def test_binary_site_store_round_trip(tmp_path):
    """`m6anet_amd pack`: data.json is parsed once into a binary site store; opening it maps the file and must hand
    back exactly what the JSON loader produces -- arrays bit for bit, ids, replicate suffixes, CSV bytes -- and it
    refuses stores packed for other normalisation factors, truncated files and foreign files."""
    import shutil
    from m6anet_amd import _io
    from m6anet_amd.__main__ import main as cli
    rep = tmp_path / "rep1"
    rep.mkdir()
    for fn in ("data.info", "data.json"):
        shutil.copyfile(os.path.join(DATA, fn), rep / fn)
    for dirs in ([DATA], [DATA, str(rep)]):
        path = str(tmp_path / ("n%d.m6astore" % len(dirs)))
        cli(["pack", "--input_dir"] + dirs + ["--out", path, "--n_processes", "2"])
        ref = data_utils.load_sites_native(dirs, 20, "norm_hct116.npz")
        st = data_utils.open_store(path, "norm_hct116.npz", 20)
        assert st.native.n_replicates == ref.native.n_replicates == len(dirs)
        for name in ("X", "site_kmers", "off", "tx_pos"):
            a, b = getattr(st, name), getattr(ref, name)
            assert a.dtype == b.dtype and np.array_equal(a, b), name
        assert np.array_equal(st.native.read_id_values, ref.native.read_id_values)
        assert np.array_equal(st.native.read_rep, ref.native.read_rep)
        assert st.tx_ids == ref.tx_ids and st.kmer5 == ref.kmer5
        g = np.random.Generator(np.random.PCG64(9))
        rp, sp, mr = g.random(ref.X.shape[0], dtype=np.float32), g.random(ref.n_sites, dtype=np.float32), g.random(ref.n_sites)
        for b, d in ((st, tmp_path / "o_store"), (ref, tmp_path / "o_json")):
            d.mkdir(exist_ok=True)
            b.native.write_csv(str(d), rp, sp, mr, write_header=True, n_threads=2)
        for fn in ("data.site_proba.csv", "data.indiv_proba.csv"):
            assert (tmp_path / "o_store" / fn).read_bytes() == (tmp_path / "o_json" / fn).read_bytes()
        with pytest.raises(ValueError, match="re-run"):
            data_utils.open_store(path, "norm_arabidopsis.npz", 20)          # packed for other norm factors
        raw = open(path, "rb").read()
        st.native.close()
        ref.native.close()
    (tmp_path / "cut.m6astore").write_bytes(raw[:len(raw) // 2])
    (tmp_path / "junk.m6astore").write_bytes(b"not a store" * 100)
    (tmp_path / "magic.m6astore").write_bytes(b"X" + raw[1:])
    for bad in ("cut", "junk", "magic", "missing"):
        with pytest.raises(_io.M6AIOError):
            data_utils.open_store(str(tmp_path / (bad + ".m6astore")))


def test_fast_16_digit_formatter_equals_printf():
    """The CSV writers print '%.16f' (inference_utils.py:62,66) through m6a_io_format_f16; it must give
    the characters printf gives, for every kind of value a probability column can hold."""
    import ctypes
    from m6anet_amd import _io
    L = _io.load()
    buf = ctypes.create_string_buffer(336)

    def fmt(v):
        n = L.m6a_io_format_f16(float(v), buf)
        out = buf.value.decode()
        assert n == len(out)
        return out

    rng = np.random.default_rng(5)
    f32 = np.concatenate([
        rng.random(200000, dtype=np.float32),                                        # probabilities
        np.exp(rng.uniform(-100, 0.5, 50000)).astype(np.float32),                    # tiny ones, down to denormal floats
        rng.integers(0, 0x7f800000, 50000, dtype=np.uint32).view(np.float32)]).astype(np.float64)   # random finite bit patterns >= 0
    f64 = np.concatenate([
        rng.random(100000), rng.integers(0, 21, 20000) / 20.0, rng.integers(0, 1001, 20000) / 1000.0,
        np.exp(rng.uniform(-800, 1.0, 50000)),
        # exact ties of the 17th digit and their neighbours: k / 2^j with short expansions
        (rng.integers(0, 2**20, 20000) * 2 + 1) / 2.0**rng.integers(1, 60, 20000)])
    special = [0.0, -0.0, 1.0, 2.0, 0.5, 1.9999999999999998, 0.99999999999999994, 5e-17, 4.9999999999999996e-17, 1.5e-16, 2.5e-16,
               5e-324, 2.2250738585072014e-308, 1e300, -0.25, 123456.789, float("nan"), float("inf"), float("-inf"),
               float(np.float32(0.033379376)), float(np.float32(1.0) - np.float32(2**-24))]
    vals = np.concatenate([f32, f64, np.array(special)])
    assert np.isfinite(vals).sum() > 500000
    bad = [(v, fmt(v), "%.16f" % v) for v in vals.tolist() if fmt(v) != "%.16f" % v]
    assert not bad, bad[:5]


def test_site_store_is_tied_to_the_content_of_its_norm_factors(tmp_path):
    """A store holds NORMALISED features: it is accepted only for the factors it was packed with -- compared by content,
    not by file name -- and a run without --norm_path (the reference then feeds un-normalised features) refuses a
    normalised store instead of silently using it; a corrupt k-mer id never reaches the GPU's embedding table."""
    import shutil
    from m6anet_amd import _io
    from m6anet_amd.constants import asset_path
    store = str(tmp_path / "b.m6astore")
    data_utils.pack_sites([DATA], store, 20, "norm_hct116.npz")
    data_utils.open_store(store, "norm_hct116.npz", 20).native.close()
    # the same factors under another name / path: accepted
    other = tmp_path / "renamed.npz"
    shutil.copyfile(asset_path("norm_hct116.npz"), other)
    data_utils.open_store(store, str(other), 20).native.close()
    # other factors under the SAME base name: refused
    d = tmp_path / "elsewhere"
    d.mkdir()
    z = np.load(asset_path("norm_hct116.npz"))
    np.savez(d / "norm_hct116.npz", kmers=z["kmers"], mean=z["mean"] + 1e-9, std=z["std"])
    with pytest.raises(ValueError, match="re-run"):
        data_utils.open_store(store, str(d / "norm_hct116.npz"), 20)
    with pytest.raises(ValueError, match="re-run"):
        data_utils.open_store(store, None, 20)                      # this run expects un-normalised features
    with pytest.raises(ValueError, match="re-run"):
        data_utils.open_store(store, "norm_hct116.npz", 30)         # another read-count filter
    # un-normalised store <-> run without factors
    raw_store = str(tmp_path / "raw.m6astore")
    data_utils.pack_sites([DATA], raw_store, 20, None)
    st = data_utils.open_store(raw_store, None, 20)
    assert np.array_equal(st.X, data_utils.load_sites([DATA], 20, None).X)
    st.native.close()
    with pytest.raises(ValueError, match="re-run"):
        data_utils.open_store(raw_store, "norm_hct116.npz", 20)
    # a k-mer id beyond the 66-word vocabulary
    nat = _io.NativeSites(store=store)
    pos = nat.site_kmers.ctypes.data - nat.off.ctypes.data            # both are views into the mapping
    nat.close()
    raw = bytearray(open(store, "rb").read())
    at = raw.find(bytes(np.asarray([0], np.int64).tobytes()), 128)      # off[0] = 0 is the first array after the header
    raw[at + pos + 5] = 200
    bad = str(tmp_path / "bad.m6astore")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(_io.M6AIOError, match="k-mer id out of range"):
        data_utils.open_store(bad, "norm_hct116.npz", 20)


def test_loader_reads_the_non_finite_literals_python_json_writes(tmp_path):
    """json.loads -- the reference's parser (data_utils.py:186) -- accepts NaN / Infinity / -Infinity, and json.dumps
    writes them: a data.json holding one must load, in the native loader as in the Python mirror, not fail the job."""
    import json
    info = open(os.path.join(DATA, "data.info")).read().splitlines()
    tx, pos, a, b, n = info[1].split(",")
    rec = json.loads(open(os.path.join(DATA, "data.json"), "rb").read()[int(a):int(b)])
    kmer, rows = next(iter(rec[tx][pos].items()))
    rows[0][0], rows[1][1], rows[2][2] = float("nan"), float("inf"), float("-inf")
    text = json.dumps(rec, separators=(",", ":")) + "\n"
    assert "NaN" in text and "Infinity" in text and "-Infinity" in text
    with open(tmp_path / "data.json", "w") as f:
        f.write(text)
    with open(tmp_path / "data.info", "w") as f:
        f.write(info[0] + "\n%s,%s,0,%d,%s\n" % (tx, pos, len(text), n))
    ref = data_utils.load_sites([str(tmp_path)], 20, "norm_hct116.npz")
    nat = data_utils.load_sites_native([str(tmp_path)], 20, "norm_hct116.npz", n_threads=2)
    assert np.array_equal(nat.X, ref.X, equal_nan=True)
    assert np.isnan(nat.X).sum() >= 1 and np.isinf(nat.X).sum() >= 2


@pytest.mark.parametrize("cuts", [[0, 101], [0, 1, 101], [0, 16, 48, 80, 101], [0, 0, 33, 33, 101, 101], [0, 7, 9, 50, 90, 101]])
def test_sharded_csv_writer_gives_the_bytes_of_one_writer(golden, tmp_path, cuts):
    """`inference --gpus N`: every rank formats the rows of ITS sites, learns their size, and pwrite()s them at its offset
    (m6a_io_csv_shard_size / m6a_io_csv_shard_write).  Whatever the cut -- empty shards included -- and whatever order the
    ranks write in, the two files are byte for byte m6a_io_write_csv's; a longer file left by an earlier run is cut to size;
    the single-sample and the replicate read-id forms both go through."""
    from m6anet_amd import _io
    for dirs in ([DATA], [DATA, DATA]):
        nat = _io.NativeSites(dirs, 20, data_utils.load_norm_factors("norm_hct116.npz"), 2)
        S, R = nat.tx_pos.size, nat.X.shape[0]
        cuts = [c * S // 101 for c in cuts]                      # 101 sites alone, 171 as two pooled replicates
        g = np.random.Generator(np.random.PCG64(5))
        rp, sp, mr = g.random(R, dtype=np.float32), g.random(S, dtype=np.float32), g.random(S)
        mr[3] = np.nan                                           # a site without reads prints 'nan'
        one = tmp_path / ("one%d" % len(dirs))
        one.mkdir(exist_ok=True)
        nat.write_csv(str(one), rp, sp, mr, write_header=True)
        want = [(one / fn).read_bytes() for fn in ("data.site_proba.csv", "data.indiv_proba.csv")]
        out = tmp_path / ("sharded%d" % len(dirs))
        out.mkdir(exist_ok=True)
        for fn in ("data.site_proba.csv", "data.indiv_proba.csv"):
            (out / fn).write_bytes(b"x" * (len(want[1]) + 12345))            # stale, longer files
        off = nat.off
        n = len(cuts) - 1
        sizes = np.array([nat.csv_shard_size(cuts[r], cuts[r + 1], rp[off[cuts[r]]:off[cuts[r + 1]]], sp[cuts[r]:cuts[r + 1]],
                                             mr[cuts[r]:cuts[r + 1]]) for r in range(n)], np.int64).reshape(n, 2)
        head = np.array(nat.csv_header_bytes(), np.int64)
        assert tuple(head + sizes.sum(axis=0)) == (len(want[0]), len(want[1]))
        for r in reversed(range(n)):                             # last rank first: rank 0 (header + final size) comes LAST
            start = head + sizes[:r].sum(axis=0)
            nat.csv_shard_write(str(out), cuts[r], cuts[r + 1], rp[off[cuts[r]]:off[cuts[r + 1]]], sp[cuts[r]:cuts[r + 1]], mr[cuts[r]:cuts[r + 1]],
                                int(start[0]), int(start[1]), header_and_totals=tuple(int(x) for x in head + sizes.sum(axis=0)) if r == 0 else None,
                                n_threads=1 + r % 3)
        assert [(out / fn).read_bytes() for fn in ("data.site_proba.csv", "data.indiv_proba.csv")] == want
        with pytest.raises(_io.M6AIOError):                      # a range outside the job
            _io._chk(_io.load().m6a_io_csv_shard_size(nat._h, None, None, None, 5, S + 1, 1, None, None))
        nat.close()


def test_sharded_csv_writer_never_writes_stale_text(tmp_path):
    """m6a_io_csv_shard_size keeps the text it formatted for the m6a_io_csv_shard_write that follows with the same range and
    arrays.  The kept text belongs to the VALUES: other values at the same addresses (a caller's buffer reused, or temporaries
    freed and reallocated between the two calls -- ADVICE r4) must be formatted again, not served from the cache."""
    from m6anet_amd import _io
    nat = _io.NativeSites([DATA], 20, data_utils.load_norm_factors("norm_hct116.npz"), 2)
    S, R = nat.tx_pos.size, nat.X.shape[0]
    g = np.random.Generator(np.random.PCG64(9))
    rp, sp, mr = g.random(R, dtype=np.float32), g.random(S, dtype=np.float32), g.random(S)
    head = nat.csv_header_bytes()

    def files(d):
        return [(d / fn).read_bytes() for fn in ("data.site_proba.csv", "data.indiv_proba.csv")]

    for what in ("same", "site value", "one read in the middle", "last read", "list input"):
        out, one = tmp_path / ("o_" + what.replace(" ", "_")), tmp_path / ("w_" + what.replace(" ", "_"))
        out.mkdir()
        one.mkdir()
        args = (rp.tolist(), sp, mr) if what == "list input" else (rp, sp, mr)      # a list: converted to a temporary per call
        sizes = nat.csv_shard_size(0, S, *args)
        if what == "site value":
            sp[S // 2] = np.float32(0.123456)
        elif what == "one read in the middle":
            rp[(R >> 1) & ~1023] = np.float32(0.5)               # an index the strided sample visits whatever the stride
        elif what == "last read":
            rp[R - 1] = np.float32(0.25)
        args = (rp.tolist(), sp, mr) if what == "list input" else (rp, sp, mr)
        nat.csv_shard_write(str(out), 0, S, *args, head[0], head[1], header_and_totals=(-1, -1))
        nat.write_csv(str(one), rp, sp, mr, write_header=True)
        got, want = files(out), files(one)
        assert got == want, what
        if what == "same":
            assert (len(got[0]), len(got[1])) == (head[0] + sizes[0], head[1] + sizes[1])
    nat.close()


def test_sharded_csv_size_call_does_not_pin_the_callers_arrays():
    """csv_shard_size keeps what the csv_shard_write that follows needs -- but only weak references to the CALLER's
    arrays (read_prob is 4 bytes per read: GBs) and strong ones to its own converted temporaries; a size call that raises, is
    never followed by its write, or is followed by another csv_* call or close() must leave nothing alive."""
    import gc
    import weakref
    from m6anet_amd import _io
    nat = _io.NativeSites([DATA], 20, data_utils.load_norm_factors("norm_hct116.npz"), 2)
    S, R = nat.tx_pos.size, nat.X.shape[0]
    g = np.random.Generator(np.random.PCG64(3))

    def arrays():
        return g.random(R, dtype=np.float32), g.random(S, dtype=np.float32), g.random(S)

    rp, sp, mr = arrays()
    refs = [weakref.ref(x) for x in (rp, sp, mr)]
    nat.csv_shard_size(0, S, rp, sp, mr)
    del rp, sp, mr
    gc.collect()
    assert all(r() is None for r in refs), "a size call without its write pinned the caller's arrays"
    # converted temporaries (float64 read_prob -> float32) are the handle's own: dropped by the next size call, write_csv or close()
    rp, sp, mr = arrays()
    nat.csv_shard_size(0, S, rp.astype(np.float64), sp, mr)
    kept = nat._shard_kept
    assert kept is not None and kept[1][0][1] is not None and kept[1][0][1].dtype == np.float32 and kept[1][1][1] is None
    nat.csv_shard_size(0, S, rp, sp, mr)
    assert nat._shard_kept is not kept and all(conv is None for _, conv in nat._shard_kept[1])
    with pytest.raises(_io.M6AIOError):
        nat.csv_shard_size(5, S + 1, rp[:0], sp[:0], mr[:0]) if False else _io._chk(_io.load().m6a_io_csv_shard_size(nat._h, None, None, None, 5, S + 1, 1, None, None))
    nat.csv_shard_size(0, S, rp, sp, mr)
    nat.close()
    assert nat._shard_kept is None
