"""Native dataprep (libm6a_io.so: m6a_io_dataprep) against the reference.

Bars:
  * eventalign.index: byte-identical to the reference's own fixture (m6anet/tests/test_dataprep.py:9-27).
  * data.info / data.json vs a reference run captured at n_processes=1 (tests/golden/dataprep_ref_run,
    made by tests/golden/make_golden.py): same records in the same order, same 7-mers, same n_reads, and
    per site the same rows BIT-EXACTLY as a multiset (the order of reads inside a site comes from an
    unstable argsort in the reference, dataprep_utils.py:444, and is machine-dependent there).
  * vs the reference's bundled data.json: np.allclose after sorting by read id -- the reference's own
    test bar (m6anet/tests/conftest.py:84-101); its bundled file was written by an older pandas and the
    current reference itself only matches it to ~4e-15.
"""
import gzip
import json
import os

import numpy as np
import pytest

from m6anet_amd import _io, data_utils

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF = os.path.join(GOLD, "ref_tests_data")


@pytest.fixture(scope="module")
def eventalign(tmp_path_factory):
    p = tmp_path_factory.mktemp("ev") / "eventalign.txt"
    p.write_bytes(gzip.open(os.path.join(REF, "eventalign.txt.gz"), "rb").read())
    return str(p)


def records(text):
    out, order = {}, []
    for line in text.splitlines():
        (tx, pp), = json.loads(line).items()
        (pos, k), = pp.items()
        (kmer, rows), = k.items()
        out[(tx, int(pos))] = (kmer, np.array(rows, dtype=np.float64))
        order.append((tx, int(pos)))
    return out, order


def sorted_rows(a):
    return a[np.lexsort(a.T[::-1])]


@pytest.mark.parametrize("tag,kw", [("msc1", dict(min_segment_count=1, compress=False)),
                                    ("msc20_compress", dict(min_segment_count=20, compress=True))])
@pytest.mark.parametrize("threads", [1, 4])
def test_dataprep_matches_reference_run(eventalign, tmp_path, tag, kw, threads):
    out = str(tmp_path / "o")
    _io.dataprep(eventalign, out, n_threads=threads, readcount_min=1, readcount_max=1000, **kw)
    assert open(os.path.join(out, "eventalign.index"), "rb").read() == open(os.path.join(REF, "eventalign.index"), "rb").read()
    got, got_order = records(open(os.path.join(out, "data.json")).read())
    want, want_order = records(gzip.open(os.path.join(GOLD, "dataprep_ref_run", tag + ".data.json.gz"), "rt").read())
    assert got_order == want_order                                   # transcripts in index order, positions ascending
    for key in want:
        assert got[key][0] == want[key][0]                           # 7-mer
        assert np.array_equal(sorted_rows(got[key][1]), sorted_rows(want[key][1])), key   # bit-exact multiset
    info = [l.split(",") for l in open(os.path.join(out, "data.info")).read().splitlines()[1:]]
    ref_info = [l.split(",") for l in open(os.path.join(GOLD, "dataprep_ref_run", tag + ".data.info")).read().splitlines()[1:]]
    assert [(r[0], r[1], r[4]) for r in info] == [(r[0], r[1], r[4]) for r in ref_info]
    blob = open(os.path.join(out, "data.json"), "rb").read()
    for tx, pos, a, b, n in info:                                     # offsets address exactly their record
        rec = json.loads(blob[int(a):int(b)])
        assert len(rec[tx][pos][got[(tx, int(pos))][0]]) == int(n)
    assert len(open(os.path.join(out, "data.log")).read().splitlines()) > 0


def test_dataprep_meets_the_reference_test_bar(eventalign, tmp_path):
    """m6anet/tests/test_dataprep.py:30-58 re-stated on the reference's bundled data.info / data.json."""
    out = str(tmp_path / "o")
    _io.dataprep(eventalign, out, readcount_min=1, readcount_max=1000, min_segment_count=1)
    got, _ = records(open(os.path.join(out, "data.json")).read())
    want, _ = records(open(os.path.join(REF, "data.json")).read())
    assert set(got) == set(want)
    for key in want:
        a, b = got[key][1], want[key][1]
        assert got[key][0] == want[key][0] and a.shape == b.shape
        ia, ib = np.argsort(a[:, -1], kind="stable"), np.argsort(b[:, -1], kind="stable")
        assert np.allclose(a[ia][:, -1], b[ib][:, -1]) and np.allclose(a[ia], b[ib])


def test_dataprep_then_loader_equals_loader_on_reference_output(eventalign, tmp_path):
    """eventalign.txt -> native dataprep -> native loader gives, per site, the same normalised
    read multiset as loading the reference's own data.json."""
    out = str(tmp_path / "o")
    _io.dataprep(eventalign, out, readcount_min=1, readcount_max=1000, min_segment_count=1)
    a = data_utils.load_sites_native([out], 20, "norm_hct116.npz")
    b = data_utils.load_sites_native([REF], 20, "norm_hct116.npz")
    ka = {(a.native.tx_id(i), int(a.tx_pos[i])): i for i in range(a.n_sites)}
    assert a.n_sites == b.n_sites == 101
    for j in range(b.n_sites):
        i = ka[(b.native.tx_id(j), int(b.tx_pos[j]))]
        xa, xb = a.X[a.off[i]:a.off[i + 1]], b.X[b.off[j]:b.off[j + 1]]
        assert np.array_equal(a.site_kmers[i], b.site_kmers[j]) and xa.shape == xb.shape
        assert np.allclose(sorted_rows(xa.astype(np.float64)), sorted_rows(xb.astype(np.float64)), rtol=1e-6, atol=1e-6)


def test_dataprep_options_and_errors(eventalign, tmp_path):
    out = str(tmp_path / "o")
    _io.dataprep(eventalign, out, readcount_min=1, readcount_max=1000, min_segment_count=20)
    n20 = len(open(os.path.join(out, "data.info")).read().splitlines()) - 1
    _io.dataprep(eventalign, out, readcount_min=1, readcount_max=1000, min_segment_count=20, skip_index=True)
    assert len(open(os.path.join(out, "data.info")).read().splitlines()) - 1 == n20 == 101
    _io.dataprep(eventalign, out, readcount_min=1, readcount_max=30, min_segment_count=20)
    assert len(open(os.path.join(out, "data.info")).read().splitlines()) - 1 < n20      # read cap bites
    with pytest.raises(_io.M6AIOError):
        _io.dataprep(str(tmp_path / "missing.txt"), out)
    with pytest.raises(_io.M6AIOError):
        _io.dataprep(eventalign, out, n_neighbors=0)
    _io.dataprep(eventalign, out, n_neighbors=2)                # the bundled file only holds 3-position runs: no 5-position window
    assert open(os.path.join(out, "data.info")).read().splitlines() == ["transcript_id,transcript_position,start,end,n_reads"]
    with pytest.raises(_io.M6AIOError):
        _io.dataprep(eventalign, str(tmp_path / "fresh"), skip_index=True)            # no index to reuse


def unpack(tmp_path, sub):
    p = tmp_path / (sub + "_eventalign.txt")
    p.write_bytes(gzip.open(os.path.join(GOLD, sub, "eventalign.txt.gz"), "rb").read())
    return str(p)


def test_parallel_index_is_stitched_at_range_boundaries(eventalign, tmp_path, monkeypatch):
    """The index is built over byte ranges on all threads and the runs that meet at a boundary are stitched
    (dataprep_utils.py:187-266 walks the file once).  The bundled file is smaller than one default range (8 MB), so the
    test shrinks the ranges: ~130 ranges of 16 KB, boundaries in the middle of reads -- same bytes as the reference's index,
    same data.json / data.info as the one-range run."""
    one = str(tmp_path / "one")
    _io.dataprep(eventalign, one, n_threads=1, readcount_min=1, readcount_max=1000, min_segment_count=20)
    monkeypatch.setenv("M6A_IO_INDEX_RANGE_KB", "16")
    for threads in (1, 3, 8):
        out = str(tmp_path / ("t%d" % threads))
        _io.dataprep(eventalign, out, n_threads=threads, readcount_min=1, readcount_max=1000, min_segment_count=20)
        assert open(os.path.join(out, "eventalign.index"), "rb").read() == open(os.path.join(REF, "eventalign.index"), "rb").read()
        for fn in ("data.json", "data.info", "data.log"):
            assert open(os.path.join(out, fn), "rb").read() == open(os.path.join(one, fn), "rb").read(), (threads, fn)


@pytest.mark.parametrize("nn", [1, 2, 3])
def test_n_neighbors_against_the_reference(tmp_path, nn):
    """--n_neighbors (m6anet/scripts/dataprep.py:45-47): windows of 2 nn + 1 consecutive positions, 3 (2 nn + 1) features and a
    (5 + 2 nn)-mer per row (roll / partition_into_continuous_positions / combine_sequence, dataprep_utils.py:51-67,117-147,
    170-183), against what the reference makes of tests/golden/dataprep_synthetic/eventalign.txt.gz (long runs with gaps,
    repeated events per position, mismatching model k-mers; captured by make_golden.py)."""
    ev = unpack(tmp_path, "dataprep_synthetic")
    out = str(tmp_path / "o")
    _io.dataprep(ev, out, n_threads=2, readcount_min=1, readcount_max=1000, min_segment_count=5, n_neighbors=nn)
    syn = os.path.join(GOLD, "dataprep_synthetic")
    assert open(os.path.join(out, "eventalign.index"), "rb").read() == open(os.path.join(syn, "eventalign.index"), "rb").read()
    got, got_order = records(open(os.path.join(out, "data.json")).read())
    want, want_order = records(gzip.open(os.path.join(syn, "nn%d.data.json.gz" % nn), "rt").read())
    assert got_order == want_order and len(want) > 3
    for key in want:
        assert got[key][0] == want[key][0] and len(want[key][0]) == 5 + 2 * nn
        assert got[key][1].shape[1] == 3 * (2 * nn + 1) + 1
        assert np.array_equal(sorted_rows(got[key][1]), sorted_rows(want[key][1])), key
    info = [l.split(",") for l in open(os.path.join(out, "data.info")).read().splitlines()[1:]]
    ref_info = [l.split(",") for l in open(os.path.join(syn, "nn%d.data.info" % nn)).read().splitlines()[1:]]
    assert [(r[0], r[1], r[4]) for r in info] == [(r[0], r[1], r[4]) for r in ref_info]


def test_read_with_non_contiguous_lines_documented_divergence(tmp_path):
    """nanopolish writes a read's events as one contiguous block; tests/golden/dataprep_noncontiguous holds the first 900
    events of the bundled file with four lines of read 82387 moved behind read 82388's block.  The reference keys its index
    rows by (contig, read_index) and adds up the line lengths of ALL rows with that key (dataprep_utils.py:187-208), so it
    writes ONE row for 82387 whose byte range has the right length but is no longer the read's lines -- it swallows the head of
    82388's block, and every later row of the chunk is still right only because the lengths add up.  Here an index row is a
    contiguous run: 82387 gets two rows, each exactly its own lines.  Pinned so the divergence stays deliberate."""
    ev = unpack(tmp_path, "dataprep_noncontiguous")
    nc = os.path.join(GOLD, "dataprep_noncontiguous")
    out = str(tmp_path / "o")
    _io.dataprep(ev, out, n_threads=2, readcount_min=1, readcount_max=1000, min_segment_count=1)
    blob = open(ev, "rb").read()
    ours = [l.split(",") for l in open(os.path.join(out, "eventalign.index")).read().splitlines()[1:]]
    ref = [l.split(",") for l in open(os.path.join(nc, "reference_index_chunk1000000.csv")).read().splitlines()[1:]]
    assert open(os.path.join(nc, "reference_index_chunk16.csv")).read() == open(os.path.join(nc, "reference_index_chunk1000000.csv")).read()

    def reads_in(row):
        return {l.split(b"\t")[3] for l in blob[int(row[2]):int(row[3])].splitlines()}

    assert all(reads_in(r) == {r[1].encode()} for r in ours)                       # every row of ours is one read's lines
    assert [r[1] for r in ours].count("82387") == 2 and [r[1] for r in ref].count("82387") == 1
    assert len(ours) == len(ref) + 1
    r87 = next(r for r in ref if r[1] == "82387")
    assert reads_in(r87) == {b"82387", b"82388"}                                  # the reference's row covers foreign lines
    assert sum(int(r[3]) - int(r[2]) for r in ours if r[1] == "82387") == int(r87[3]) - int(r87[2])
    # every row that does not touch the two reads is the same in both
    assert [r for r in ours if r[1] not in ("82387", "82388")] == [r for r in ref if r[1] not in ("82387", "82388")]


def test_float_text_is_pythons_repr():
    """data.json holds repr(float) (json.dump in the reference, dataprep_utils.py:473-480): 400 000 doubles -- random bit
    patterns, decimal-looking values of every magnitude from 1e-320 to 1e300, integers around 2**53, the 1e-4 / 1e16
    boundaries of repr's two notations, signed zeros, non-finite values in json's spelling."""
    import ctypes as C
    L = _io.load()
    buf = C.create_string_buffer(40)
    g = np.random.Generator(np.random.PCG64(1))
    vals = list(g.integers(0, 2**64 - 1, size=150_000, dtype=np.uint64).view(np.float64))
    vals += list(np.round(g.normal(100, 30, 60_000), 2)) + list(np.round(g.random(60_000) * 0.02, 5)) + list(g.random(30_000) * 10.0 ** g.integers(-320, 300, 30_000))
    vals += list(g.integers(-2**60, 2**60, 50_000).astype(np.float64)) + [float(2**53 + k) for k in range(-5, 6)]
    for e in range(-6, 18):
        for m in (1.0, 0.9999999999999999, 1.0000000000000002, 9.999999999999999, 5.0, 1.5):
            vals += [m * 10.0 ** e, -m * 10.0 ** e]
    vals += [0.0, -0.0, 1e-4, 9.999e-5, 1e16, 9999999999999998.0, 123456789012345.6, 5e-324, 1.7976931348623157e308]
    bad = []
    for v in vals:
        v = float(v)
        n = L.m6a_io_py_repr(v, buf)
        want = repr(v) if np.isfinite(v) else ("NaN" if np.isnan(v) else ("Infinity" if v > 0 else "-Infinity"))
        if n < 0 or buf.value.decode() != want:
            bad.append((v.hex(), buf.value.decode(), want))
    assert not bad, bad[:5]


def test_rounded_numbers_fast_path_is_pythons_repr():
    """Round 6: a third of data.json's numbers are means rounded to one decimal (all of them have three with --compress); the
    writer prints those as k / 10^d with trailing zeros dropped instead of searching for the shortest digits.  Wherever the fast
    path accepts a value its text must be Python's repr of it; where it declines (zero, huge, values that are not np.round
    results) the general m6a_io_py_repr path is pinned by the test above."""
    import ctypes as C
    L = _io.load()
    buf = C.create_string_buffer(40)
    g = np.random.Generator(np.random.PCG64(7))
    accepted = 0
    for digits in (1, 3):
        xs = np.concatenate([g.normal(100, 40, 120_000), g.random(60_000) * 0.05, g.random(40_000) * 10.0 ** g.integers(-6, 10, 40_000),
                             -g.random(20_000) * 300, np.arange(-2000, 2000) / 8.0, [0.05, 0.15, 0.25, 0.35, 1e8 + 0.1, 999999999.9, 1e9, 1e12 + 0.5, 0.0, -0.0, 1e-7]])
        for v in np.round(xs, digits):
            v = float(v)
            n = L.m6a_io_repr_rounded(v, digits, buf)
            if n < 0:
                assert v == 0 or abs(v) >= 1e9 or round(v, digits) != v, v       # declines only where it says it does
                continue
            accepted += 1
            assert buf.value.decode() == repr(v), (v.hex(), digits, buf.value.decode(), repr(v))
        for v in (g.random(20_000) * 100):                                        # NOT rounded: must decline (or, by luck, be exact)
            n = L.m6a_io_repr_rounded(float(v), digits, buf)
            assert n < 0 or buf.value.decode() == repr(float(v))
    assert accepted > 400_000


def test_workers_run_ahead_by_bytes_and_write_in_order(tmp_path, monkeypatch):
    """Transcripts are processed on all threads and written in index order as soon as their predecessors are; how far the
    workers run ahead is a budget of finished-but-unwritten JSON.  With the budget at its minimum (1 MB against ~12 MB of
    output) and 8 threads the workers block and resume constantly: the three files are those of one thread."""
    text = gzip.open(os.path.join(REF, "eventalign.txt.gz"), "rt").read()
    header, body = text.split("\n", 1)
    big = tmp_path / "big.txt"
    with open(big, "w") as f:
        f.write(header + "\n")
        for k in range(30):
            f.write(body.replace("ENST", "C%dENST" % k) if k else body)
    one = str(tmp_path / "one")
    _io.dataprep(str(big), one, n_threads=1, readcount_min=1, readcount_max=1000, min_segment_count=5)
    assert os.path.getsize(os.path.join(one, "data.json")) > 8 << 20
    monkeypatch.setenv("M6A_IO_PENDING_MB", "1")
    monkeypatch.setenv("M6A_IO_INDEX_RANGE_KB", "256")
    out = str(tmp_path / "t8")
    _io.dataprep(str(big), out, n_threads=8, readcount_min=1, readcount_max=1000, min_segment_count=5)
    for fn in ("eventalign.index", "data.json", "data.info", "data.log"):
        assert open(os.path.join(out, fn), "rb").read() == open(os.path.join(one, fn), "rb").read(), fn


def test_line_scanner_extra_columns_and_a_file_that_ends_on_a_page_boundary(tmp_path):
    """Round 6's one-pass field scanner reads 16 bytes at a time: it must never load across the end of the mapping (a file whose
    size is a multiple of the page size has NOTHING mapped behind its last byte), must ignore columns beyond the 15 the reference
    reads (nanopolish --samples / --signal-index add some), and must cope with lines of any length: the outputs are those of the
    plain file."""
    import mmap
    text = gzip.open(os.path.join(REF, "eventalign.txt.gz"), "rt").read()
    header, body = text.split("\n", 1)
    lines = body.rstrip("\n").split("\n")[:4000]
    plain = tmp_path / "plain.txt"
    plain.write_text(header + "\n" + "\n".join(lines) + "\n")
    want = str(tmp_path / "want")
    _io.dataprep(str(plain), want, n_threads=2, readcount_min=1, readcount_max=1000, min_segment_count=1)
    assert os.path.getsize(os.path.join(want, "data.json")) > 10_000
    page = mmap.PAGESIZE
    for tag, with_newline in (("page_nl", True), ("page_nonl", False)):
        g = np.random.Generator(np.random.PCG64(5))
        wide = [l + "\t" + "s" * int(g.integers(0, 40)) + ("\textra" if i % 3 == 0 else "") for i, l in enumerate(lines)]
        blob = (header + "\tsamples\n" + "\n".join(wide)).encode()
        tail = b"\n" if with_newline else b""
        pad = (-(len(blob) + len(tail))) % page                  # grow the LAST line's extra column until the file ends on a page boundary
        blob += b"x" * pad + tail
        assert len(blob) % page == 0
        f = tmp_path / (tag + ".txt")
        f.write_bytes(blob)
        out = str(tmp_path / tag)
        _io.dataprep(str(f), out, n_threads=3, readcount_min=1, readcount_max=1000, min_segment_count=1)
        for fn in ("data.json", "data.info"):
            assert open(os.path.join(out, fn), "rb").read() == open(os.path.join(want, fn), "rb").read(), (tag, fn)


def test_dataprep_edge_files(tmp_path):
    """Header only; a last line without its newline; CRLF line ends; a single read: the indexer's byte ranges stay exact and
    nothing is lost or invented."""
    text = gzip.open(os.path.join(REF, "eventalign.txt.gz"), "rt").read()
    header, body = text.split("\n", 1)
    lines = body.rstrip("\n").split("\n")
    # header only
    p = tmp_path / "h.txt"
    p.write_text(header + "\n")
    out = str(tmp_path / "h")
    _io.dataprep(str(p), out, n_threads=2)
    assert open(os.path.join(out, "eventalign.index")).read() == "transcript_id,read_index,pos_start,pos_end\n"
    assert open(os.path.join(out, "data.info")).read() == "transcript_id,transcript_position,start,end,n_reads\n"
    assert open(os.path.join(out, "data.json")).read() == ""
    # the first 600 events, with and without the final newline, and with CRLF: same index rows (CRLF: longer lines), same sites
    base = header + "\n" + "\n".join(lines[:600])
    res = {}
    for tag, txt in (("nl", base + "\n"), ("nonl", base), ("crlf", (base + "\n").replace("\n", "\r\n"))):
        p = tmp_path / (tag + ".txt")
        p.write_bytes(txt.encode())
        out = str(tmp_path / tag)
        _io.dataprep(str(p), out, n_threads=3, readcount_min=1, readcount_max=1000, min_segment_count=1)
        idx = [l.split(",") for l in open(os.path.join(out, "eventalign.index")).read().splitlines()[1:]]
        blob = p.read_bytes()
        for tx, read, a, b in idx:                                   # every row is exactly one read's whole lines
            seg = blob[int(a):int(b)]
            assert seg and all(l.split(b"\t")[0].decode() == tx and l.split(b"\t")[3].decode() == read for l in seg.splitlines())
        assert int(idx[-1][3]) == len(blob) and int(idx[0][2]) == len(header) + (2 if tag == "crlf" else 1)
        res[tag] = (len(idx), records(open(os.path.join(out, "data.json")).read()))
    assert res["nl"][0] == res["nonl"][0] == res["crlf"][0] > 5
    for tag in ("nonl", "crlf"):
        got, want = res[tag][1], res["nl"][1]
        assert got[1] == want[1] and all(np.array_equal(got[0][k][1], want[0][k][1]) for k in want[0])


@pytest.fixture(scope="module")
def io_lib_with_test_hooks(tmp_path_factory):
    """The fault-injection hook (M6A_IO_TEST_THROW) is compiled only into test builds (-DM6A_IO_TEST_HOOKS; ADVICE r5: the shipped
    libm6a_io.so reads no such variable).  tests/sanitize.sh hands over its own sanitized hook build in M6A_IO_LIB_HOOKS."""
    import subprocess
    given = os.environ.get("M6A_IO_LIB_HOOKS")
    if given:
        return given
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path_factory.mktemp("iolib") / "libm6a_io_hooks.so")
    subprocess.check_call([os.environ.get("CXX", "g++"), "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", "-DM6A_IO_TEST_HOOKS",
                           "-I" + os.path.join(repo, "include"), "-I" + os.path.join(repo, "m6anet_amd", "csrc"),
                           os.path.join(repo, "m6anet_amd", "csrc", "m6a_io.cpp"), "-o", out])
    return out


def test_shipped_io_library_has_no_fault_injection_hook():
    assert b"M6A_IO_TEST_THROW" not in open(_io.LIB_PATH, "rb").read() or "M6A_IO_LIB" in os.environ


@pytest.mark.parametrize("phase", ["index", "transcript", "index_file", "bookkeeping"])
def test_out_of_memory_on_any_thread_is_an_error_code_not_an_abort(eventalign, tmp_path, phase, io_lib_with_test_hooks):
    """m6a_io_dataprep runs its phases on std::threads; an exception on a worker would be std::terminate, one on the
    calling thread would unwind through the C ABI.  M6A_IO_TEST_THROW makes the named phase throw std::bad_alloc once (the
    index pass on a pool thread, the transcript pass on a worker, the index file on its background writer, `bookkeeping` on the
    CALLING thread right after the index file was opened): the call must return M6A_IO_ENOMEM, the process must live, no
    half-written output may stay behind (streams and descriptors are closed by guards, partial files removed), and
    the library must work again afterwards."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, os; sys.path.insert(0, %r)\n"
            "from m6anet_amd import _io\n"
            "try:\n"
            "    _io.dataprep(%r, %r, n_threads=4)\n"
            "    print('NO ERROR')\n"
            "except _io.M6AIOError as e:\n"
            "    print('RC', e.code)\n"
            "print('FDS', len(os.listdir('/proc/self/fd')))\n" % (repo, eventalign, str(tmp_path)))
    outputs = ("data.json", "data.info", "data.log", "eventalign.index")
    env = dict(os.environ, M6A_IO_TEST_THROW=phase, M6A_IO_INDEX_RANGE_KB="16", M6A_IO_LIB=io_lib_with_test_hooks)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-500:])           # alive: no std::terminate, no abort
    lines = r.stdout.split()
    assert lines[:2] == ["RC", "-2"], (r.stdout, r.stderr[-500:])       # M6A_IO_ENOMEM
    fds_after_failure = int(lines[3])
    assert [f for f in outputs if os.path.exists(os.path.join(str(tmp_path), f))] == [], phase   # nothing half-written stays
    env.pop("M6A_IO_TEST_THROW")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    lines = r.stdout.split()
    assert r.returncode == 0 and lines[:2] == ["NO", "ERROR"], (r.stdout, r.stderr[-500:])
    assert all(os.path.getsize(os.path.join(str(tmp_path), f)) > 0 for f in outputs if f != "data.log")
    assert fds_after_failure == int(lines[3]), "a failed dataprep leaked descriptors"     # same count as after a clean run
