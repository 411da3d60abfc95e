"""GPU parity tests: the HIP path (through the C ABI, via m6anet_amd.engine) against the CPU
oracle and the golden vectors captured from the reference.

Tolerances (stated per assertion):
  * read probabilities: rtol=1e-5, atol=1e-8 -- the reference's own bar for
    data.indiv_proba.csv (m6anet/tests/test_inference.py:32, np.allclose defaults);
  * site probabilities from identical read probabilities: BIT-exact (np.array_equal) -- indices are
    exact replays of the NumPy stream, the 20-term product runs left to right like np.prod and the
    mean follows ndarray.mean's float32 pairwise tree; north_star asks for 1e-5;
  * mod_ratio, flush groups, which sites get which draws: exact.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from m6anet_amd import synthetic                      # noqa: E402
from m6anet_amd.constants import DEFAULT_READ_THRESHOLD  # noqa: E402

THR = np.float32(DEFAULT_READ_THRESHOLD)


def same_sites(got, want):
    """Site probabilities are BIT-identical to the reference's whenever the read probabilities are:
    the kernels replay np.prod's left-to-right float32 product and ndarray.mean's pairwise float32
    sum (8 accumulator chains per <=128-element leaf, NumPy's split points) exactly."""
    return np.array_equal(got, want, equal_nan=True)



@pytest.fixture(scope="module")
def orc():
    from oracle import m6a_oracle
    m6a_oracle.build()
    return m6a_oracle


@pytest.fixture(scope="module")
def engines(weights):
    from m6anet_amd.engine import M6ANetEngine
    return {name: M6ANetEngine(weights=w) for name, w in weights.items()}


@pytest.fixture(scope="module")
def eng(engines):
    return engines["hct116"]


def rand_sites(seed, n_reads_per_site):
    g = np.random.Generator(np.random.PCG64(seed))
    n = np.asarray(n_reads_per_site, np.int64)
    off = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    X = np.clip(g.standard_normal((int(off[-1]), 9)), -6, 6).astype(np.float32)
    km = g.integers(0, 66, size=(len(n), 3)).astype(np.uint8)
    return X, km, off


# uniform bags: both pooling kernels (m6a_set_table_variant: 1 = LDS gather, 2 = bags in registers)
TABLE_VARIANTS = ((1, "table"), (2, "table-reg"))


class table_variant:
    def __init__(self, eng, mode):
        self.eng, self.mode = eng, mode

    def __enter__(self):
        self.eng.set_table_variant(self.mode)

    def __exit__(self, *exc):
        self.eng.set_table_variant(0)


def rand_probs(seed, off):
    g = np.random.Generator(np.random.PCG64(seed))
    # skewed like real read probabilities: mostly small, some large
    return (g.random(int(off[-1]), dtype=np.float32) ** 4).astype(np.float32)


# ------------------------------------------------------------------ the checker, checked where the kernels are checked ----
def test_oracle_pinning_tests_pass_on_this_box_too():
    """the driver's GPU run selects `-m gpu`, which deselects every test that holds the ORACLE to the
    reference's captured vectors (tests/test_oracle_golden.py, tests/test_reference_at_scale.py: 53 CPU tests, 12 s) -- so the
    evidence that the checker equals the reference lived in builder / judge runs only.  This gpu-marked test runs that subset
    as it is, on the box whose HIP-vs-oracle results it vouches for (another CPU, another libm: the oracle is -ffp-contract=off
    C with its own expf, so its bits must not depend on either)."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_oracle_golden.py", "tests/test_reference_at_scale.py", "-q", "-m", "not gpu",
                        "-p", "no:cacheprovider"], cwd=repo, capture_output=True, text=True, timeout=900)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0 and " passed" in tail and "failed" not in tail, (r.stdout[-1500:], r.stderr[-500:])
    assert int(tail.split(" passed")[0].split()[-1]) >= 53, tail


# ------------------------------------------------------------------ encoder -----------------
def test_encoder_golden_all_models(golden, engines):
    b = golden("bundled_inputs.npz")
    want = golden("bundled_readprob.npz")
    for name, e in engines.items():
        got = e.get_read_probability(b["X"], b["site_kmers"], b["off"])
        assert np.allclose(got, want[name], rtol=1e-5, atol=1e-8), name


@pytest.mark.parametrize("bags", [
    [1], [20], [31], [32], [33], [1, 1, 1, 1, 1], [20] * 7, [0, 5, 0, 0, 7, 0], [3] * 100,
    [1] * 200 + [40] + [1] * 50, [662, 20, 21, 500, 33], list(range(1, 70)),
])
def test_encoder_vs_oracle_shapes(eng, orc, weights, bags):
    """The library's automatic choice (nothing set: what the CLI, INTEGRATION.md's stub and bench.py's headline run) is the
    16-slot arithmetic on every input -- the oracle's BITS, not a tolerance."""
    X, km, off = rand_sites(len(bags) * 7 + 1, bags)
    got = eng.get_read_probability(X, km, off)
    assert eng.last_encoder_variant == "general16"
    assert eng.last_encoder_kernel == ("enc_site16_kernel" if min(bags) >= 16 else "enc_kernel")
    want = orc.encode_reads(weights["hct116"], X, km, off)
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_encoder_ragged_large_vs_oracle(engines, orc, weights):
    g = np.random.Generator(np.random.PCG64(5))
    bags = g.integers(20, 300, size=3000)
    X, km, off = rand_sites(11, bags)
    for name in ("hek293t_glori", "arabidopsis"):
        got = engines[name].get_read_probability(X, km, off)
        want = orc.encode_reads(weights[name], X, km, off, n_threads=8)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), name          # auto = the oracle's bits
        engines[name].set_encoder_variant(4)                                            # the opt-in 12-slot kernel: the reference test's bar
        try:
            fast = engines[name].get_read_probability(X, km, off)
            assert engines[name].last_encoder_kernel == "enc_csite_kernel"
        finally:
            engines[name].set_encoder_variant(0)
        assert np.allclose(fast, want, rtol=1e-5, atol=1e-8), name


def test_encoder_extreme_inputs(eng, orc, weights):
    X, km, off = rand_sites(3, [64] * 4)
    X[:64] = 6.0
    X[64:128] = -6.0
    X[128:192] = 0.0
    got = eng.get_read_probability(X, km, off)
    want = orc.encode_reads(weights["hct116"], X, km, off)
    assert np.all(np.isfinite(got))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("bags", [[1], [20] * 64, [3] * 100, list(range(1, 70)), [662, 20, 21, 500, 33], [0, 5, 0, 0, 7, 0],
                                  [16, 17, 40, 16, 700, 16, 16, 33]])
def test_encoder_general16_is_the_oracle_bit_for_bit(engines, orc, weights, bags):
    """The 16-slot kernel and the oracle perform the same float32 operations in the same order from the first feature to
    the division of the sigmoid (both restate the reference: DESIGN.md section 2): whatever the bags, the read
    probabilities must be the same BITS -- ordinary features, features at the clip, and far outside it."""
    X, km, off = rand_sites(len(bags) * 3 + 2, bags)
    if X.shape[0] > 40:
        X[5] = 6.0
        X[6] = -6.0
        X[7] = 0.0
        X[8, :] = [40.0, -35.0, 12.0, 0.5, -0.25, 3.0, -60.0, 1e-3, 9.0]
    for model in ("hct116", "hek293t_glori", "arabidopsis", "hek293t_m6ace"):
        e = engines[model]
        e.set_encoder_variant(1)
        try:
            got = e.get_read_probability(X, km, off)
        finally:
            e.set_encoder_variant(0)
        want = orc.encode_reads(weights[model], X, km, off)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (model, int((got.view(np.uint32) != want.view(np.uint32)).sum()))


@pytest.mark.parametrize("bags", [[16] * 50, [20] * 333, [16, 17, 40, 16, 700, 16, 16, 33], list(range(16, 120)),
                                  [32] * 9 + [31] * 9, [1000, 16, 16, 16, 2000], [20] * 7 + [19], [16], [47]])
def test_encoder_site16_scalar_chain_is_the_walk_and_the_oracle_bit_for_bit(engines, orc, weights, bags):
    """Bags >= 16 reads: the 16-slot arithmetic runs behind the 12-slot kernel's scalar 32-bit site chain
    (enc_site16_kernel) instead of the per-lane walk (enc_kernel, mode 3).  Same float32 operations: the read
    probabilities must be the same BITS as the walk's and the oracle's, NaN and far-out features included."""
    X, km, off = rand_sites(sum(bags) % 89 + 3, bags)
    if X.shape[0] > 40:
        X[5] = 6.0
        X[6] = -6.0
        X[8, :] = [40.0, -35.0, 12.0, 0.5, -0.25, 3.0, -60.0, 1e-3, 9.0]
        X[9, 2] = np.nan
        X[10] = 1e6
    for model in ("hct116", "hek293t_glori"):
        e = engines[model]
        try:
            e.set_encoder_variant(1)
            got = e.get_read_probability(X, km, off)
            assert e.last_encoder_variant == "general16" and e.last_encoder_kernel == "enc_site16_kernel"
            e.set_encoder_variant(3)
            walk = e.get_read_probability(X, km, off)
            assert e.last_encoder_variant == "general16" and e.last_encoder_kernel == "enc_kernel"
        finally:
            e.set_encoder_variant(0)
        want = orc.encode_reads(weights[model], X, km, off)
        assert np.array_equal(got.view(np.uint32), walk.view(np.uint32)), model
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), model


def test_encoder_site16_needs_bags_of_16(eng):
    """One bag below 16 reads anywhere in the call: variant 1 falls back to the walk by itself."""
    X, km, off = rand_sites(2, [20] * 40 + [15])
    eng.set_encoder_variant(1)
    try:
        eng.get_read_probability(X, km, off)
        assert eng.last_encoder_kernel == "enc_kernel"
        X, km, off = rand_sites(2, [20] * 40 + [16])
        eng.get_read_probability(X, km, off)
        assert eng.last_encoder_kernel == "enc_site16_kernel"
    finally:
        eng.set_encoder_variant(0)


@pytest.mark.parametrize("variant", [0, 1, 2, 4])
def test_encoder_nan_and_huge_features(eng, orc, weights, variant):
    """Layer 1's ReLU is the clamp modifier of the batch-norm fma (m6a_kernels.hip bn_relu): a NaN feature must still come
    out as a NaN probability for THAT read only (the reference propagates it; the hardware's default clamp would have
    turned it into 0), and features far outside any normalised signal (|x| = 1e6: activations ~1e7, nowhere near the 2^64
    the clamp saturates at) must match the oracle like any other."""
    X, km, off = rand_sites(41, [20] * 30)
    X[7, 3] = np.nan
    X[300, 0] = np.nan
    X[100] = 1e6
    X[101] = -1e6
    X[102, 4] = 3e5
    eng.set_encoder_variant(variant)
    try:
        got = eng.get_read_probability(X, km, off)
    finally:
        eng.set_encoder_variant(0)
    want = orc.encode_reads(weights["hct116"], X, km, off)
    bad = np.zeros(got.size, bool)
    bad[[7, 300]] = True
    assert np.isnan(got[bad]).all() and np.isnan(want[bad]).all()
    assert np.isfinite(got[~bad]).all()
    assert np.allclose(got[~bad], want[~bad], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("variant,name", [(1, "general16"), (2, "csite12")])
@pytest.mark.parametrize("bags", [[16] * 50, [20] * 333, [16, 17, 40, 16, 700, 16, 16, 33], list(range(16, 120)),
                                  [32] * 9 + [31] * 9, [1000, 16, 16, 16, 2000]])
def test_encoder_variants_vs_oracle(engines, orc, weights, variant, name, bags):
    """Both layer-1 formulations (16 K-slots; 12 K-slots with per-site constants folded) on bags of
    >= 16 reads, all four checkpoints' topology via two of them."""
    X, km, off = rand_sites(sum(bags) % 97 + variant, bags)
    for model in ("hct116", "hek293t_m6ace"):
        e = engines[model]
        e.set_encoder_variant(variant)
        try:
            got = e.get_read_probability(X, km, off)
            assert e.last_encoder_variant == name
        finally:
            e.set_encoder_variant(0)
        want = orc.encode_reads(weights[model], X, km, off)
        assert np.allclose(got, want, rtol=1e-5, atol=1e-8), (model, name)


def test_encoder_variant_from_the_environment(weights, monkeypatch):
    """M6A_ENCODER preselects the kernel for callers that cannot call m6a_set_encoder_variant; a value the library does not
    know makes m6a_create FAIL (a typo must not silently select another kernel)."""
    from m6anet_amd._lib import M6AError
    from m6anet_amd.engine import M6ANetEngine
    X, km, off = rand_sites(5, [20] * 40)
    for value, want in (("general16", "enc_site16_kernel"), ("csite12", "enc_csite_kernel"), ("fast", "enc_csite_kernel"),
                        ("walk16", "enc_kernel"), ("reference", "enc_site16_kernel"), ("auto", "enc_site16_kernel"), ("", "enc_site16_kernel")):
        monkeypatch.setenv("M6A_ENCODER", value)
        e = M6ANetEngine(weights=weights["hct116"])
        try:
            e.get_read_probability(X, km, off)
            assert e.last_encoder_kernel == want, value
        finally:
            e.close()
    for value in ("nonsense", "general", "General16", "csite"):
        monkeypatch.setenv("M6A_ENCODER", value)
        with pytest.raises(M6AError, match="M6A_ENCODER"):
            M6ANetEngine(weights=weights["hct116"])
    monkeypatch.delenv("M6A_ENCODER")
    e = M6ANetEngine(weights=weights["hct116"])
    try:
        e.get_read_probability(X, km, off)
        assert e.last_encoder_kernel == "enc_site16_kernel"
    finally:
        e.close()


def test_encoder_variant_selection_and_precondition(eng):
    from m6anet_amd._lib import M6AError
    X, km, off = rand_sites(1, [20] * 40)
    eng.get_read_probability(X, km, off)
    assert eng.last_encoder_variant == "general16" and eng.last_encoder_kernel == "enc_site16_kernel"    # auto: the reference's bits
    X15, km15, off15 = rand_sites(2, [20] * 40 + [15])
    eng.get_read_probability(X15, km15, off15)
    assert eng.last_encoder_variant == "general16" and eng.last_encoder_kernel == "enc_kernel"   # one bag of 15 reads: the walk
    eng.set_encoder_variant(4)                                # fast: the 12-slot kernel where it applies, 16 slots elsewhere
    try:
        eng.get_read_probability(X, km, off)
        assert eng.last_encoder_variant == "csite12" and eng.last_encoder_kernel == "enc_csite_kernel"
        eng.get_read_probability(X15, km15, off15)
        assert eng.last_encoder_variant == "general16" and eng.last_encoder_kernel == "enc_kernel"
    finally:
        eng.set_encoder_variant(0)
    X, km, off = rand_sites(3, [3] * 200)
    eng.set_encoder_variant(2)                                # forcing the 12-slot kernel on small bags
    try:
        with pytest.raises(M6AError):
            eng.get_read_probability(X, km, off)              # is reported, not silently wrong
    finally:
        eng.set_encoder_variant(0)
    got = eng.get_read_probability(X, km, off)                # and the context keeps working
    assert np.all(np.isfinite(got))


def test_encoder_empty(eng):
    out = eng.get_read_probability(np.zeros((0, 9), np.float32), np.zeros((0, 3), np.uint8), np.zeros(1, np.int64))
    assert out.shape == (0,)
    out = eng.get_read_probability(np.zeros((0, 9), np.float32), np.zeros((3, 3), np.uint8), np.zeros(4, np.int64))
    assert out.shape == (0,)


def test_bag_forward_golden(eng, golden):
    g = golden("bag_forward.npz")
    got = eng.forward(g["X"], g["kmer"], bag=20)
    assert np.allclose(got, g["site_prob"], rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------ site pooling -------------
CASES = [(5, 16, 2, 0), (100, 16, 2, 0), (1000, 16, 2, 0), (50, 8, 3, 0), (20, 13, 2, 7), (30, 51, 2, 0)]


@pytest.mark.parametrize("T,bs,spb,seed", CASES)
def test_pool_bundled_golden(eng, golden, T, bs, spb, seed):
    """Ragged real bags (20..662 reads) against full reference runs -- the scan kernels (101 sites do not pay for 60
    index tables), or the index-table kernel once earlier calls with the same (seed, T) have."""
    b = golden("bundled_inputs.npz")
    p = golden("bundled_readprob.npz")["hct116"]
    g = golden("bundled_site.npz")
    key = f"T{T}_bs{bs}_spb{spb}_seed{seed}"
    site, mod = eng.calculate_site_proba(p, b["off"], T, 20, THR, seed, bs, spb)
    assert eng.last_pool_variant.startswith("scan") or eng.last_pool_variant == "ragged-table"
    assert same_sites(site, g[key + "_site"])
    assert np.array_equal(mod, g[key + "_mod"])


@pytest.mark.parametrize("T", [100, 1000])
def test_pool_synthetic_golden(eng, golden, T):
    g = golden("synthetic_small.npz")
    for tag, mode, variant in (("uniform20", 1, "table"), ("uniform20", 2, "table-reg"), ("ragged", 0, "scan")):
        with table_variant(eng, mode):
            site, mod = eng.calculate_site_proba(g[f"{tag}_readprob"], g[f"{tag}_off"], T, 20, THR)
        assert eng.last_pool_variant.startswith(variant) and (mode == 0 or eng.last_pool_variant == variant)
        assert same_sites(site, g[f"{tag}_site_T{T}"]), (tag, variant)
        assert np.array_equal(mod, g[f"{tag}_mod"]), (tag, variant)


@pytest.mark.parametrize("n", [1, 2, 3, 19, 20, 21, 31, 32])
@pytest.mark.parametrize("T", [1, 63, 64, 65, 200])
def test_pool_uniform_table_vs_oracle(eng, orc, n, T):
    S = 131
    off = np.arange(S + 1, dtype=np.int64) * n
    p = rand_probs(n * 1000 + T, off)
    want_site, want_mod = orc.site_pool(p, off, T, THR, seed=3)
    for mode, name in TABLE_VARIANTS:
        with table_variant(eng, mode):
            site, mod = eng.calculate_site_proba(p, off, T, 20, THR, seed=3)
        assert eng.last_pool_variant == name
        assert same_sites(site, want_site), name
        assert np.array_equal(mod, want_mod), name


@pytest.mark.parametrize("n", [4, 8, 12, 16, 20, 24, 28, 32])
@pytest.mark.parametrize("shift", [0, 1, 2, 3])
def test_pool_register_kernel_quad_loads_at_any_dword_alignment(eng, orc, n, shift):
    """pool_reg_kernel loads bags of n % 4 == 0 reads with global_load_dwordx4; the read-probability array a caller hands
    over is only dword-aligned in general (a view into a larger device buffer, a rank's slice of a job): every alignment
    mod 16 bytes, device pointers, against the oracle bit for bit."""
    import torch
    S = 700
    off = np.arange(S + 1, dtype=np.int64) * n
    p = rand_probs(n * 10 + shift, off)
    want_site, want_mod = orc.site_pool(p, off, 130, THR, seed=1)
    buf = torch.zeros(p.size + 8, dtype=torch.float32, device="cuda")
    view = buf[shift:shift + p.size]
    view.copy_(torch.from_numpy(p))
    assert view.data_ptr() % 16 == (buf.data_ptr() + 4 * shift) % 16
    with table_variant(eng, 2):
        site, mod = eng.calculate_site_proba(view, torch.from_numpy(off).cuda(), 130, 20, THR, seed=1)
    assert eng.last_pool_variant == "table-reg"
    assert same_sites(site.cpu().numpy(), want_site) and np.array_equal(mod.cpu().numpy(), want_mod)


@pytest.mark.parametrize("bags", [
    [33] * 40, [64] * 35, [65] * 33, [1000] * 3, [1024, 1025, 1500, 20], [4096, 20, 4097, 30], [3000, 25] * 20 + [4096], [20, 21] * 30,
    [1, 2, 1, 20, 1, 40], list(range(20, 84)), [2048, 20, 4097],
    [0, 25, 0, 0, 1, 40, 0], [0] * 5 + [1024] + [0] * 3 + [2, 3] * 20,
])
def test_pool_scan_vs_oracle(eng, orc, bags):
    off = np.concatenate([[0], np.cumsum(bags)]).astype(np.int64)
    p = rand_probs(len(bags), off)
    try:
        for T in (7, 100):
            want_site, want_mod = orc.site_pool(p, off, T, THR, seed=11)
            for driver, name in ((1, "scan-group"), (2, "scan-site"), (3, "ragged-table"), (0, None)):
                eng.set_scan_driver(driver)
                if driver == 3 and max(bags) > 4096:          # index tables exist for bags of <= 4096 reads
                    with pytest.raises(Exception, match="M6A_EUNSUPPORTED"):
                        eng.calculate_site_proba(p, off, T, 20, THR, seed=11)
                    continue
                site, mod = eng.calculate_site_proba(p, off, T, 20, THR, seed=11)
                assert eng.last_pool_variant == (name or eng.last_pool_variant)
                assert eng.last_pool_variant.startswith("scan") or eng.last_pool_variant == "ragged-table"
                assert same_sites(site, want_site), (T, driver)
                assert np.array_equal(mod, want_mod, equal_nan=True)
    finally:
        eng.set_scan_driver(0)


@pytest.mark.parametrize("K", [1, 5, 19, 21, 64])
def test_pool_other_sample_counts(eng, orc, K):
    bags = [20, 25, 30, 100, 20, 33] * 6
    off = np.concatenate([[0], np.cumsum(bags)]).astype(np.int64)
    p = rand_probs(K, off)
    want_site, want_mod = orc.site_pool(p, off, 50, THR, seed=5, n_samples=K)
    try:
        for driver in (1, 2, 3):
            eng.set_scan_driver(driver)
            site, mod = eng.calculate_site_proba(p, off, 50, K, THR, seed=5)
            assert same_sites(site, want_site), driver
            assert np.array_equal(mod, want_mod)
    finally:
        eng.set_scan_driver(0)


@pytest.mark.parametrize("T", [1, 2, 7, 8, 9, 15, 16, 17, 120, 127, 128, 129, 135, 136, 137, 255, 256, 257, 264,
                               999, 1001, 1003, 2047, 2049, 4099])
def test_pool_mean_is_numpy_pairwise_sum(eng, orc, T):
    """Every shape of NumPy's pairwise-sum tree: a single short leaf (T < 8: all tail), one leaf with
    and without a tail, the first split (129), leaves of unequal length, tails on the last leaf of a
    deep tree; through the table kernel (uniform bags, n = 1 included) and both scan drivers."""
    cases = [([20] * 37, "table"), ([1] * 9, "table"), ([32] * 11, "table"),
             ([20, 33, 1, 64, 47, 2, 21] * 5, "scan")]
    for bags, variant in cases:
        off = np.concatenate([[0], np.cumsum(bags)]).astype(np.int64)
        p = rand_probs(T + len(bags), off)
        want_site, want_mod = orc.site_pool(p, off, T, THR, seed=6)
        # table: LDS / register kernel; scan: per group / per site / per-bag-size index tables
        for driver in (1, 2) if variant == "table" else (1, 2, 3):
            eng.set_scan_driver(driver)
            eng.set_table_variant(min(driver, 2))
            try:
                site, mod = eng.calculate_site_proba(p, off, T, 20, THR, seed=6)
            finally:
                eng.set_scan_driver(0)
                eng.set_table_variant(0)
            assert eng.last_pool_variant.startswith(variant if driver < 3 else "ragged-table")
            if variant == "table":
                assert eng.last_pool_variant == TABLE_VARIANTS[driver - 1][1]
            assert same_sites(site, want_site), (T, bags[:3], driver, np.abs(site - want_site).max())
            assert np.array_equal(mod, want_mod)


@pytest.mark.parametrize("bs,spb", [(16, 2), (1, 2), (7, 3), (64, 2), (16, 1), (5, 5)])
def test_pool_group_geometry(eng, orc, bs, spb):
    S = 203
    for n, mode in ((20, 1), (20, 2), (None, 0)):
        bags = [20] * S if n else list(np.random.Generator(np.random.PCG64(S)).integers(20, 60, size=S))
        off = np.concatenate([[0], np.cumsum(bags)]).astype(np.int64)
        p = rand_probs(bs * 10 + spb, off)
        with table_variant(eng, mode):
            site, mod = eng.calculate_site_proba(p, off, 30, 20, THR, seed=9, batch_size=bs, save_per_batch=spb)
        want_site, _ = orc.site_pool(p, off, 30, THR, seed=9, batch_size=bs, save_per_batch=spb)
        assert same_sites(site, want_site), (bs, spb, n, mode)


@pytest.mark.parametrize("T", [1024, 1025, 1500, 3000, 10000, 16385, 40003])
def test_pool_long_iterations(eng, orc, T):
    """num_iterations beyond one 16-round LDS chunk of the index row (table kernel, T > 1024) and
    long replays in the scan kernels; 10000 is what the reference's own test uses.  Beyond 16384 the
    pairwise-sum tree is deeper than the register kernel's 8-entry stack: it must hand over to the LDS kernel."""
    S = 70 if T < 10000 else 40 if T < 16000 else 10
    for bags, mode, variant in (([20] * S, 1, "table"), ([20] * S, 2, "table"), ([20, 27, 33, 64, 100] * (S // 5), 0, "scan"),
                                ([20, 27, 33, 64, 100] * (S // 5), 3, "ragged-table")):
        off = np.concatenate([[0], np.cumsum(bags)]).astype(np.int64)
        p = rand_probs(T, off)
        eng.set_scan_driver(mode if variant == "ragged-table" else 1 if variant == "scan" else 0)
        try:
            with table_variant(eng, mode if variant == "table" else 0):
                site, mod = eng.calculate_site_proba(p, off, T, 20, THR, seed=2)
        finally:
            eng.set_scan_driver(0)
        assert eng.last_pool_variant.startswith(variant)
        want_site, want_mod = orc.site_pool(p, off, T, THR, seed=2, n_threads=8)
        assert same_sites(site, want_site), (T, variant, mode, eng.last_pool_variant)
        assert np.array_equal(mod, want_mod)


def test_pool_many_small_groups_and_one_huge_group(eng, orc):
    off = np.arange(0, 20 * 3001, 20, dtype=np.int64)
    p = rand_probs(77, off)
    for bs, spb in ((1, 2), (4096, 2), (3000, 1)):       # 1-site groups; one flush group holding every site
        want_site, _ = orc.site_pool(p, off, 12, THR, seed=4, batch_size=bs, save_per_batch=spb, n_threads=8)
        for mode, _name in TABLE_VARIANTS:
            with table_variant(eng, mode):
                site, mod = eng.calculate_site_proba(p, off, 12, 20, THR, seed=4, batch_size=bs, save_per_batch=spb)
            assert same_sites(site, want_site), (bs, spb, mode, eng.last_pool_variant)


def test_pool_uniform_random_configurations(eng, orc):
    """Random (bag size, num_iterations, batch geometry, seed) for uniform bags: the register kernel, the LDS
    kernel and the oracle must agree bit for bit (tails, several leaves, many positions per group, ...)."""
    g = np.random.Generator(np.random.PCG64(2026))
    for _ in range(24):
        n = int(g.integers(1, 33))
        T = int(g.choice([1, 3, 8, 9, 40, 129, 250, 777, 1000, 1023, 1500, 2600]))
        bs = int(g.choice([1, 3, 16, 50]))
        spb = int(g.choice([1, 2, 3]))
        S = int(g.integers(1, 400))
        seed = int(g.integers(0, 2 ** 32))
        off = np.arange(S + 1, dtype=np.int64) * n
        p = rand_probs(seed % 1000, off)
        want_site, want_mod = orc.site_pool(p, off, T, THR, seed=seed, batch_size=bs, save_per_batch=spb, n_threads=8)
        for mode, name in TABLE_VARIANTS:
            with table_variant(eng, mode):
                site, mod = eng.calculate_site_proba(p, off, T, 20, THR, seed=seed, batch_size=bs, save_per_batch=spb)
            assert eng.last_pool_variant == name
            assert same_sites(site, want_site), (n, T, bs, spb, S, seed, name)
            assert np.array_equal(mod, want_mod), (n, T, bs, spb, S, seed, name)


def test_pool_ragged_random_configurations(eng, orc):
    """Random ragged jobs (bag sizes from 1 to a few thousand, mixed), iteration counts and batch geometry:
    both scan drivers and the index-table kernel against the oracle, bit for bit."""
    g = np.random.Generator(np.random.PCG64(77))
    for _ in range(12):
        S = int(g.integers(2, 120))
        kind = int(g.integers(0, 3))
        if kind == 0:
            bags = g.integers(1, 40, size=S)
        elif kind == 1:
            bags = g.integers(20, 700, size=S)
        else:
            bags = np.where(g.random(S) < 0.1, g.integers(1000, 3000, size=S), g.integers(20, 64, size=S))
        bags[int(g.integers(0, S))] += 1                                   # never uniform
        T = int(g.choice([1, 5, 8, 33, 130, 257, 1000]))
        bs, spb = int(g.choice([1, 4, 16])), int(g.choice([1, 2, 3]))
        seed = int(g.integers(0, 2 ** 32))
        off = np.concatenate([[0], np.cumsum(bags)]).astype(np.int64)
        p = rand_probs(seed % 997, off)
        want_site, want_mod = orc.site_pool(p, off, T, THR, seed=seed, batch_size=bs, save_per_batch=spb, n_threads=8)
        try:
            for driver in (1, 2, 3):
                if driver == 3 and bags.max() > 4096:
                    continue
                eng.set_scan_driver(driver)
                site, mod = eng.calculate_site_proba(p, off, T, 20, THR, seed=seed, batch_size=bs, save_per_batch=spb)
                assert eng.last_pool_variant.startswith("scan" if driver < 3 else "ragged-table")
                assert same_sites(site, want_site), (S, kind, T, bs, spb, seed, driver)
                assert np.array_equal(mod, want_mod), (S, kind, T, bs, spb, seed, driver)
        finally:
            eng.set_scan_driver(0)


def test_pool_index_table_cache(eng, orc):
    """The per-bag-size index tables are kept across calls: new bag sizes are added (the arena grows past its
    first 32 slots), a new seed / iteration count / longer stream rebuilds them, and coming back gives the
    same bits.  Every call is checked against the oracle."""
    g = np.random.Generator(np.random.PCG64(5))

    def run(bags, T, seed, bs=16, spb=2):
        off = np.concatenate([[0], np.cumsum(bags)]).astype(np.int64)
        p = rand_probs(int(off[-1]) % 1000, off)
        site, mod = eng.calculate_site_proba(p, off, T, 20, THR, seed=seed, batch_size=bs, save_per_batch=spb)
        assert eng.last_pool_variant == "ragged-table"
        want_site, want_mod = orc.site_pool(p, off, T, THR, seed=seed, batch_size=bs, save_per_batch=spb, n_threads=8)
        assert same_sites(site, want_site), (len(bags), T, seed, bs, spb)
        assert np.array_equal(mod, want_mod)
        return site

    eng.set_scan_driver(3)
    try:
        few = list(g.integers(20, 30, size=150))                 # <= 10 bag sizes
        many = list(g.integers(20, 140, size=400))               # ~120 bag sizes: the arena must grow, old slots survive
        a = run(few, 100, 0)
        run(many, 100, 0)
        assert np.array_equal(run(few, 100, 0), a)
        run(few + [1, 1, 2, 1024, 3], 100, 0)                    # bags of one read (no table), the largest size
        run(many, 100, 12345)                                    # new seed: new stream, new tables
        run(many, 37, 12345)                                     # new T*K
        run(few, 100, 0, bs=40, spb=3)                           # larger flush groups: a longer stream
        assert np.array_equal(run(few, 100, 0), a)
    finally:
        eng.set_scan_driver(0)


def test_pool_auto_picks_tables_for_large_ragged_jobs(eng, orc):
    bags = list(np.random.Generator(np.random.PCG64(8)).integers(20, 60, size=3000))
    off = np.concatenate([[0], np.cumsum(bags)]).astype(np.int64)
    p = rand_probs(3, off)
    site, mod = eng.calculate_site_proba(p, off, 64, 20, THR, seed=21)
    assert eng.last_pool_variant == "ragged-table"               # 3000 sites, 40 bag sizes: tables pay
    want_site, want_mod = orc.site_pool(p, off, 64, THR, seed=21, n_threads=8)
    assert same_sites(site, want_site) and np.array_equal(mod, want_mod)
    site2, _ = eng.calculate_site_proba(p[:int(off[40])], off[:41], 64, 20, THR, seed=99)
    assert eng.last_pool_variant.startswith("scan")              # 40 sites, new seed: replay the stream instead
    assert same_sites(site2, orc.site_pool(p[:int(off[40])], off[:41], 64, THR, seed=99)[0])


def test_pool_flush_group_at_a_time_earns_its_tables(eng, orc):
    """The reference-side stub hands over ONE flush group per call.  No single call pays for the
    index tables it lacks, so the first ones replay the stream (scan kernels); the sites pooled that way are credited,
    and once they would have paid for the missing tables those are built and every later call is a table gather.
    Every call, on either kernel, equals the oracle bit for bit."""
    bags = np.random.Generator(np.random.PCG64(31)).integers(20, 50, size=32 * 40)
    off = np.concatenate([[0], np.cumsum(bags)]).astype(np.int64)
    p = rand_probs(9, off)
    seen = []
    for g0 in range(0, len(bags), 32):
        sl = slice(int(off[g0]), int(off[g0 + 32]))
        o = off[g0:g0 + 33] - off[g0]
        site, mod = eng.calculate_site_proba(p[sl], o, 200, 20, THR, seed=4242, batch_size=32)
        seen.append(eng.last_pool_variant)
        want_site, want_mod = orc.site_pool(p[sl], o, 200, THR, seed=4242, batch_size=32)
        assert same_sites(site, want_site) and np.array_equal(mod, want_mod), (g0, seen[-1])
    assert seen[0].startswith("scan") and seen[-1] == "ragged-table"
    first_table = seen.index("ragged-table")
    assert 5 <= first_table <= 20 and all(v == "ragged-table" for v in seen[first_table + 8:])


def test_pool_seeds_differ_and_repeat(eng):
    off = np.arange(101, dtype=np.int64) * 20
    p = rand_probs(1, off)
    a, _ = eng.calculate_site_proba(p, off, 100, seed=0)
    b, _ = eng.calculate_site_proba(p, off, 100, seed=1)
    c, _ = eng.calculate_site_proba(p, off, 100, seed=0)
    assert np.array_equal(a, c)
    assert not np.array_equal(a, b)


def test_bad_arguments(eng):
    from m6anet_amd._lib import M6AError
    off = np.arange(3, dtype=np.int64) * 20
    p = np.zeros(40, np.float32)
    with pytest.raises(M6AError):
        eng.calculate_site_proba(p, off, 0)
    with pytest.raises(M6AError):
        eng.calculate_site_proba(p, off, 10, n_samples=65)
    with pytest.raises(M6AError):
        eng.calculate_site_proba(p, off, 10, batch_size=0)
    with pytest.raises(M6AError):
        eng.calculate_site_proba(p, np.array([0, 30, 20], np.int64), 10)
    # rng_mode: M6A_RNG_NUMPY (0) is the only value the three signatures take (include/m6a.h): anything else is a
    # bad ARGUMENT (M6A_EINVAL = -1), on every entry point that carries the parameter, and leaves the context usable
    import ctypes as C
    L, h = eng._L, eng._h
    site, mod = np.empty(2, np.float32), np.empty(2, np.float64)
    X, km = np.zeros((40, 9), np.float32), np.zeros((2, 3), np.uint8)
    for bad in (1, -1, 7):
        assert L.m6a_site_pool(h, p.ctypes.data, off.ctypes.data, 2, 10, 20, C.c_float(0.5), 0, bad, 16, 2, site.ctypes.data, mod.ctypes.data) == -1
        assert b"M6A_RNG_NUMPY" in L.m6a_last_error(h)
        assert L.m6a_infer(h, X.ctypes.data, km.ctypes.data, off.ctypes.data, 2, 10, 20, C.c_float(0.5), 0, bad, 16, 2, None, site.ctypes.data,
                           mod.ctypes.data) == -1
        assert L.m6a_job_begin(h, 10, 20, C.c_float(0.5), 0, bad, 16, 2, 0, 0) == -1
    assert L.m6a_site_pool(h, p.ctypes.data, off.ctypes.data, 2, 10, 20, C.c_float(0.5), 0, 0, 16, 2, site.ctypes.data, mod.ctypes.data) == 0
    for knob in (eng.set_table_variant, eng.set_scan_driver, eng.set_encoder_variant):
        with pytest.raises(M6AError):
            knob(7)
        with pytest.raises(M6AError):
            knob(-1)
    with pytest.raises(M6AError):
        eng.validate_pool(p, off, 0)
    # the 12-slot encoder forced onto bags it cannot take: reported at the next sync, not a wrong answer
    d = synthetic.make_sites(50, 5, seed=3)
    eng.set_encoder_variant(2)
    try:
        with pytest.raises(M6AError):
            eng.get_read_probability(d["X"], d["site_kmers"], d["off"])
    finally:
        eng.set_encoder_variant(0)
    assert np.all(np.isfinite(eng.get_read_probability(d["X"], d["site_kmers"], d["off"])))


# ------------------------------------------------------------------ end to end -----------------
def test_infer_end_to_end_vs_oracle(eng, orc, weights):
    d = synthetic.make_sites(20000, 20, seed=99)
    rp, site, mod = eng.infer(d["X"], d["site_kmers"], d["off"], 1000)
    p = orc.encode_reads(weights["hct116"], d["X"], d["site_kmers"], d["off"], n_threads=8)
    assert np.allclose(rp, p, rtol=1e-5, atol=1e-8)
    want_site, want_mod = orc.site_pool(rp, d["off"], 1000, THR, n_threads=8)
    assert same_sites(site, want_site)
    assert np.array_equal(mod, want_mod)
    # and against the oracle's own read probabilities: north_star's 1e-5
    o_site, _ = orc.site_pool(p, d["off"], 1000, THR, n_threads=8)
    assert np.abs(site - o_site).max() <= 1e-5


@pytest.mark.parametrize("driver", [1, 2, 3])
def test_infer_ragged_end_to_end_vs_oracle(engines, orc, weights, driver):
    d = synthetic.make_sites(600, (50, 500), seed=5)
    engines["hek293t_glori"].set_scan_driver(driver)
    try:
        rp, site, mod = engines["hek293t_glori"].infer(d["X"], d["site_kmers"], d["off"], 1000)
    finally:
        engines["hek293t_glori"].set_scan_driver(0)
    p = orc.encode_reads(weights["hek293t_glori"], d["X"], d["site_kmers"], d["off"], n_threads=8)
    assert np.allclose(rp, p, rtol=1e-5, atol=1e-8)
    want_site, want_mod = orc.site_pool(rp, d["off"], 1000, THR, n_threads=8)
    assert same_sites(site, want_site)
    assert np.array_equal(mod, want_mod)


# ------------------------------------------------------------------ validation-style forward -----
@pytest.mark.parametrize("key,seed,T,same_batching", [("seed0_T5_bs16", 0, 5, True), ("seed7_T3_bs101", 7, 3, True),
                                                     ("seed1_T12_bs1", 1, 12, False)])
def test_validate_pool_vs_reference_validate(eng, golden, key, seed, T, same_batching):
    """`validate` at num_workers=0 (training_utils.py:213-268) -- sampler without
    replacement, per-pass predictions, float32 mean -- from the reference's own read probabilities."""
    g = golden("validate.npz")
    b = golden("bundled_inputs.npz")
    p = golden("bundled_readprob.npz")["hct116"]
    y, avg = eng.validate_pool(p, b["off"], T, seed=seed)
    if same_batching:
        assert np.array_equal(y, g[key + "_y_pred"]) and np.array_equal(avg, g[key + "_y_pred_avg"])
    else:   # batch_size 1: the reference's own read probabilities differ by 1 ulp between batchings
        assert np.abs(y - g[key + "_y_pred"]).max() <= 2.4e-7 and np.abs(avg - g[key + "_y_pred_avg"]).max() <= 2.4e-7


def test_validate_forward_vs_oracle_and_errors(engines, orc, weights):
    from m6anet_amd._lib import M6AError
    e = engines["hek293t_glori"]
    d = synthetic.make_sites(3000, (20, 300), seed=12)
    y, avg, rp = e.validate_forward(d["X"], d["site_kmers"], d["off"], n_iterations=4, seed=5, want_read_probs=True)
    p = orc.encode_reads(weights["hek293t_glori"], d["X"], d["site_kmers"], d["off"], n_threads=8)
    assert np.allclose(rp, p, rtol=1e-5, atol=1e-8)
    want_y, want_avg = orc.validate(rp, d["off"], 4, seed=5)
    assert np.array_equal(y, want_y) and np.array_equal(avg, want_avg)
    # device tensors give the same bits
    import torch
    dev = torch.device("cuda:0")
    ty, tavg = e.validate_forward(*(torch.from_numpy(d[k]).to(dev) for k in ("X", "site_kmers", "off")), n_iterations=4, seed=5)
    assert np.array_equal(ty.cpu().numpy(), y) and np.array_equal(tavg.cpu().numpy(), avg)
    # other sample counts; a bag shorter than the sample is NumPy's ValueError
    y7, avg7 = e.validate_pool(rp, d["off"], 3, n_samples=7, seed=1)
    w7, wavg7 = orc.validate(rp, d["off"], 3, seed=1, k=7)
    assert np.array_equal(y7, w7) and np.array_equal(avg7, wavg7)
    with pytest.raises(M6AError):
        e.validate_pool(rp[:44], np.array([0, 25, 44], np.int64), 2)


def test_validate_dictionary_like_the_reference(eng, golden):
    from m6anet_amd import training_utils as tu
    g = golden("validate.npz")
    b = golden("bundled_inputs.npz")
    res = tu.validate(eng, b["X"], b["site_kmers"], b["off"], g["y_true"], n_iterations=5, seed=0)
    assert set(res) == {"y_pred", "y_true", "compute_time", "roc_auc", "pr_auc", "avg_loss"}
    assert len(res["y_pred"]) == 5 and np.abs(np.asarray(res["y_pred"]) - g["seed0_T5_bs16_y_pred"]).max() <= 1e-5
    assert abs(res["avg_loss"] - float(g["seed0_T5_bs16_avg_loss"])) < 1e-4
    assert abs(res["roc_auc"] - float(g["seed0_T5_bs16_roc_auc"])) < 1e-3


def test_device_tensors_match_host_path(eng):
    import torch
    d = synthetic.make_sites(5000, 20, seed=3)
    rp_h, site_h, mod_h = eng.infer(d["X"], d["site_kmers"], d["off"], 100)
    dev = torch.device("cuda:0")
    X = torch.from_numpy(d["X"]).to(dev)
    km = torch.from_numpy(d["site_kmers"]).to(dev)
    off = torch.from_numpy(d["off"]).to(dev)
    eng.use_torch_stream()
    try:
        rp_d, site_d, mod_d = eng.infer(X, km, off, 100)
        eng.sync()
        assert np.array_equal(rp_d.cpu().numpy(), rp_h)
        assert np.array_equal(site_d.cpu().numpy(), site_h)
        assert np.array_equal(mod_d.cpu().numpy(), mod_h)
        _, site_n, _ = eng.infer(X, km, off, 100, want_read_probs=False)
        eng.sync()
        assert np.array_equal(site_n.cpu().numpy(), site_h)
    finally:
        eng.set_stream(None)


def test_job_offset_shards_equal_whole_job(eng):
    """Multi-GPU rule on one GPU: shards cut by shard_plan and run with set_job_offset reproduce the
    unsharded job bit for bit (uniform bags -> table kernel, ragged -> scan kernel)."""
    from m6anet_amd.engine import shard_plan
    from m6anet_amd._lib import M6AError
    for bag in (20, (20, 90)):
        d = synthetic.make_sites(3000, bag, seed=8)
        rp, site, mod = eng.infer(d["X"], d["site_kmers"], d["off"], 64)
        cuts = shard_plan(d["off"], 3)
        parts = []
        try:
            for r in range(3):
                a, b = int(cuts[r]), int(cuts[r + 1])
                eng.set_job_offset(a)
                off = d["off"][a:b + 1] - d["off"][a]
                X = d["X"][d["off"][a]:d["off"][b]]
                parts.append(eng.infer(X, d["site_kmers"][a:b], off, 64))
            eng.set_job_offset(16)      # inside a group for the default geometry? 16 starts group 1: fine
            eng.infer(d["X"][:20 * 0 + d["off"][32]], d["site_kmers"][:32], d["off"][:33], 8)
            eng.set_job_offset(32)      # batch 2 continues the group {1,2}: must be refused
            with pytest.raises(M6AError):
                eng.infer(d["X"][:d["off"][32]], d["site_kmers"][:32], d["off"][:33], 8)
        finally:
            eng.set_job_offset(0)
        assert np.array_equal(np.concatenate([p[0] for p in parts]), rp)
        assert np.array_equal(np.concatenate([p[1] for p in parts]), site)
        assert np.array_equal(np.concatenate([p[2] for p in parts]), mod)


def test_soak_repeated_runs_are_bit_identical(engines):
    """40 back-to-back runs of both pooling paths and both encoder kernels under a busy GPU: any
    missing LDS ordering / prefetch hazard shows up as a run that differs."""
    import torch
    dev = torch.device("cuda:0")
    for bag, model in ((20, "hct116"), ((20, 200), "hek293t_glori")):
        eng = engines[model]
        d = synthetic.make_sites(30000, bag, seed=12)
        X, km, off = (torch.from_numpy(d[k]).to(dev) for k in ("X", "site_kmers", "off"))
        eng.use_torch_stream()
        try:
            ref = None
            for i in range(40):
                eng.set_encoder_variant((2 if i & 1 else 0) if bag == 20 else 0)
                rp, site, mod = eng.infer(X, km, off, 300)
                eng.sync()
                cur = (rp.clone(), site.clone(), mod.clone())
                key = i & 1 if bag == 20 else 0
                if ref is None:
                    ref = {}
                if key not in ref:
                    ref[key] = cur
                else:
                    assert all(torch.equal(a, b) for a, b in zip(cur, ref[key])), (bag, i)
            if bag == 20:       # the two encoder kernels agree to float32 rounding
                assert torch.allclose(ref[0][0], ref[1][0], rtol=1e-5, atol=1e-8)
        finally:
            eng.set_encoder_variant(0)
            eng.set_stream(None)


@pytest.mark.parametrize("workload,sites", [("uniform", 3001), ("ragged", 700)])
def test_bench_self_launch_two_ranks_equal_the_whole_job(workload, sites):
    """`python bench.py --gpus 2` with no launcher starts its own two ranks (here on gloo, sharing the one GPU of a
    test box; the 8-GPU run is the same code on backend nccl): every rank runs the HIP engine on its flush-group-
    aligned shard with the job offset, one gather brings site_prob / mod_ratio to rank 0, and --verify recomputes
    the whole job unsharded on rank 0's GPU: bit-identical."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, M6A_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--workload", workload, "--sites", str(sites),
                          "--iters", "60", "--steps", "2", "--warmup", "1", "--min-seconds", "0", "--verify", "--no-cpu-baseline"],
                         capture_output=True, text=True, env=env, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["verify"] is True
    assert d["value"] > 0 and d["config"]["sites_per_gpu"] == sites


def test_host_ring_with_empty_and_tiny_bags(engines, orc, weights):
    """The chunked host-pointer path on a job with empty sites (no reads at all), one-read bags and runs of empty
    sites at chunk boundaries: same results as the device path, NaN site probability / mod_ratio for empty sites
    (the reference would divide by zero there too)."""
    import torch
    eng = engines["hct116"]
    eng.prepare_host_io()
    g = np.random.Generator(np.random.PCG64(21))
    bags = g.integers(0, 60, size=90_000)
    bags[g.random(bags.size) < 0.2] = 0
    bags[40_000:40_500] = 0
    X, km, off = rand_sites(5, bags)                                  # 2.1 M reads = 76 MB: four chunks
    dev = torch.device("cuda:0")
    rp_d, site_d, mod_d = eng.infer(torch.from_numpy(X).to(dev), torch.from_numpy(km).to(dev), torch.from_numpy(off).to(dev), 40)
    eng.sync()
    rp, site, mod = eng.infer(X, km, off, 40)
    assert np.array_equal(rp, rp_d.cpu().numpy())
    assert np.array_equal(site, site_d.cpu().numpy(), equal_nan=True) and np.array_equal(mod, mod_d.cpu().numpy(), equal_nan=True)
    assert np.isnan(site[bags == 0]).all() and np.isnan(mod[bags == 0]).all() and np.isfinite(site[bags > 0]).all()
    p = orc.encode_reads(weights["hct116"], X, km, off, n_threads=8)
    assert np.allclose(rp, p, rtol=1e-5, atol=1e-8)
    want_site, want_mod = orc.site_pool(rp, off, 40, THR, n_threads=8)
    assert same_sites(site, want_site) and np.array_equal(mod, want_mod, equal_nan=True)


def test_native_rccl_gather_single_rank(eng):
    """m6a_comm_unique_id / m6a_comm_init / m6a_gather (include/m6a.h) on the one GPU a test box has: a
    communicator of one rank still goes through ncclCommInitRank and the grouped ncclSend/ncclRecv (to itself)
    on the context's stream.  World sizes > 1 cannot share a GPU under RCCL; bench.py --gpus N drives them."""
    import torch
    from m6anet_amd.engine import comm_unique_id
    ident = comm_unique_id()
    assert len(ident) == 128
    eng.comm_init(ident, 0, 1)
    try:
        dev = torch.device("cuda:0")
        site = torch.rand(100_003, dtype=torch.float32, device=dev)
        mod = torch.rand(100_003, dtype=torch.float64, device=dev)
        for _ in range(2):
            s_all, m_all = eng.gather(site, mod, np.array([0, 100_003]), dst=0)
            eng.sync()
            assert torch.equal(s_all, site) and torch.equal(m_all, mod)
        with pytest.raises(Exception, match="M6A_EINVAL"):
            eng.comm_init(ident, 0, 1)                       # one communicator per context
    finally:
        eng.comm_destroy()


def test_host_pointer_pipeline_matches_device_path(engines, orc, weights):
    """Host buffers go through the pinned staging ring in chunks (H2D of chunk k+1 under the encoder of chunk k);
    the results must be those of the device-pointer path bit for bit -- ragged bags so that chunk boundaries fall
    inside flush groups, a job large enough for several ring wrap-arounds, and read probabilities optional."""
    import torch
    eng = engines["hek293t_glori"]
    eng.prepare_host_io()                                          # jobs under ~200 MB only use the ring once it exists
    d = synthetic.make_sites(60_000, (20, 90), seed=11)            # 3.3 M reads = 119 MB of X: 5 chunks of 24 MB
    dev = torch.device("cuda:0")
    tX, tk, to = (torch.from_numpy(d[k]).to(dev) for k in ("X", "site_kmers", "off"))
    rp_d, site_d, mod_d = eng.infer(tX, tk, to, 50)
    eng.sync()
    for want_rp in (True, False):
        rp, site, mod = eng.infer(d["X"], d["site_kmers"], d["off"], 50, want_read_probs=want_rp)
        assert np.array_equal(site, site_d.cpu().numpy()) and np.array_equal(mod, mod_d.cpu().numpy())
        if want_rp:
            assert np.array_equal(rp, rp_d.cpu().numpy())
    rp2 = eng.get_read_probability(d["X"], d["site_kmers"], d["off"])
    assert np.array_equal(rp2, rp_d.cpu().numpy())
    # against the oracle on a slice that spans the first chunk boundary (699 050 reads per 24 MB slot)
    cut = int(np.searchsorted(d["off"], 699_050))
    a, b = cut - 40, cut + 40
    sl = slice(int(d["off"][a]), int(d["off"][b]))
    p = orc.encode_reads(weights["hek293t_glori"], d["X"][sl], d["site_kmers"][a:b], d["off"][a:b + 1] - d["off"][a])
    assert np.allclose(rp2[sl], p, rtol=1e-5, atol=1e-8)


def test_pinned_host_buffers_skip_the_staging_copy_same_bits(engines):
    """Caller buffers that are already page-locked (hipHostMalloc / torch's pin_memory()) are DMA'd chunk by chunk straight
    from / into the caller's memory (m6a_host_ring.hip is_pinned_host): same chunks, same kernels, same bits as the pageable
    path and the device path -- inputs pinned, outputs pinned, both, and the encoder-only entry point."""
    import torch
    eng = engines["hek293t_glori"]
    eng.prepare_host_io()
    d = synthetic.make_sites(60_000, (20, 90), seed=12)            # 119 MB of X: 5 chunks
    R, S = int(d["off"][-1]), 60_000
    want = eng.infer(d["X"], d["site_kmers"], d["off"], 50)        # pageable path (pinned against the device path above)
    pin = tuple(torch.from_numpy(d[k]).pin_memory() for k in ("X", "site_kmers", "off"))

    def pinned_outs():
        return (torch.full((R,), -1.0, dtype=torch.float32).pin_memory(), torch.full((S,), -1.0, dtype=torch.float32).pin_memory(),
                torch.full((S,), -1.0, dtype=torch.float64).pin_memory())
    for ins, outs in ((pin, None), ((d["X"], d["site_kmers"], d["off"]), pinned_outs()), (pin, pinned_outs())):
        got = eng.infer(*ins, 50, out=outs)
        for g, w in zip(got, want):
            g = g.numpy() if hasattr(g, "numpy") and not isinstance(g, np.ndarray) else g
            assert np.array_equal(g, w)
    rp = torch.empty(R, dtype=torch.float32).pin_memory()
    eng.get_read_probability(*pin, out=rp)
    assert np.array_equal(rp.numpy(), want[0])
    # a pinned array that is reused right after the call returns: the call must have finished reading it
    Xp = pin[0].clone().pin_memory()
    got = eng.infer(Xp, pin[1], pin[2], 50)
    Xp.zero_()
    eng.sync()
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


def test_pinned_numpy_arrays_without_torch(engines):
    """engine.pinned_empty (m6a_host_alloc): page-locked NumPy arrays for callers without torch -- the same in-place DMA path, the
    same bits; freeing follows the array's lifetime."""
    from m6anet_amd.engine import pinned_empty
    eng = engines["hct116"]
    eng.prepare_host_io()
    d = synthetic.make_sites(40_000, (20, 60), seed=13)
    R, S = int(d["off"][-1]), 40_000
    want = eng.infer(d["X"], d["site_kmers"], d["off"], 30)
    X, km, off = pinned_empty((R, 9), np.float32), pinned_empty((S, 3), np.uint8), pinned_empty(S + 1, np.int64)
    X[:], km[:], off[:] = d["X"], d["site_kmers"], d["off"]
    outs = (pinned_empty(R, np.float32), pinned_empty(S, np.float32), pinned_empty(S, np.float64))
    for o in outs:
        o[:] = -1
    got = eng.infer(X, km, off, 30, out=outs)
    assert all(np.array_equal(g, w) for g, w in zip(got, want))
    view = outs[0][10:20]
    del outs, got                                                   # the view keeps the allocation alive
    assert np.array_equal(view, want[0][10:20])
    assert pinned_empty(0, np.float32).size == 0


def test_host_is_pinned_checks_the_whole_range(eng):
    """m6a_host_is_pinned = the test the host-pointer calls apply before they DMA caller memory in place: the WHOLE
    range must lie inside ONE page-locked allocation -- not just its first and last byte."""
    import ctypes as C
    import torch
    from m6anet_amd.engine import pinned_empty
    L = eng._L
    a, b = pinned_empty(1 << 20, np.uint8), pinned_empty(1 << 20, np.uint8)
    pa, pb = a.ctypes.data, b.ctypes.data
    assert L.m6a_host_is_pinned(pa, 1 << 20) == 1 and L.m6a_host_is_pinned(pa + 4096, (1 << 20) - 4096) == 1
    assert L.m6a_host_is_pinned(pa + 12345, 1000) == 1
    assert L.m6a_host_is_pinned(pa, 1 << 30) == 0                       # runs out of its allocation
    lo, hi = min(pa, pb), max(pa, pb)
    assert L.m6a_host_is_pinned(lo, hi - lo + (1 << 20)) == 0           # first and last byte page-locked, two allocations
    page = np.zeros(1 << 20, np.uint8)
    assert L.m6a_host_is_pinned(page.ctypes.data, page.nbytes) == 0     # pageable
    t = torch.empty(1 << 18, dtype=torch.float32).pin_memory()
    assert L.m6a_host_is_pinned(t.data_ptr(), t.numel() * 4) == 1       # torch's pinned allocator (hipHostMalloc blocks, sub-allocated)
    dev = torch.empty(1024, dtype=torch.float32, device="cuda")
    assert L.m6a_host_is_pinned(dev.data_ptr(), 4096) == 0              # device memory is not host memory
    assert L.m6a_host_is_pinned(None, 10) == 0 and L.m6a_host_is_pinned(pa, 0) == 0
    # hipHostRegister'ed pageable memory: one registration is in; a range over two registrations with a pageable page between is out
    hip = C.CDLL("libamdhip64.so")
    big = np.zeros(3 << 20, np.uint8)
    base = (big.ctypes.data + 4095) & ~4095
    assert hip.hipHostRegister(C.c_void_p(base), C.c_size_t(1 << 20), 0) == 0
    assert hip.hipHostRegister(C.c_void_p(base + (1 << 20) + 4096), C.c_size_t(1 << 20), 0) == 0
    try:
        assert L.m6a_host_is_pinned(base, 1 << 20) == 1
        assert L.m6a_host_is_pinned(base + (1 << 20) + 4096, 1 << 20) == 1
        assert L.m6a_host_is_pinned(base, (2 << 20) + 4096) == 0
    finally:
        hip.hipHostUnregister(C.c_void_p(base))
        hip.hipHostUnregister(C.c_void_p(base + (1 << 20) + 4096))


def test_infer_equals_encode_then_pool_for_every_pooling_kernel(engines):
    """m6a_infer sets the pooling up on a side stream while the encoder runs (a dry launch_pool); whatever kernel the
    pooling takes -- forced scan drivers, index tables, both uniform-bag kernels -- the fused call must give what
    m6a_encode_reads followed by m6a_site_pool gives, bit for bit, also when the kernel choice changes between calls."""
    import torch
    eng = engines["hek293t_glori"]
    dev = torch.device("cuda:0")
    try:
        for bags, knobs in (((20, 120), [("scan", 1), ("scan", 2), ("scan", 3), ("scan", 0)]),
                            (20, [("table", 1), ("table", 2), ("table", 0)])):
            d = synthetic.make_sites(1500, bags, seed=21)
            tX, tk, to = (torch.from_numpy(d[k]).to(dev) for k in ("X", "site_kmers", "off"))
            for kind, v in knobs:
                (eng.set_scan_driver if kind == "scan" else eng.set_table_variant)(v)
                for T in (30, 64):
                    rp, site, mod = eng.infer(tX, tk, to, T, seed=T)
                    rp2 = eng.get_read_probability(tX, tk, to)
                    site2, mod2 = eng.calculate_site_proba(rp2, to, T, seed=T)
                    eng.sync()
                    assert np.array_equal(rp.cpu().numpy(), rp2.cpu().numpy())
                    assert np.array_equal(site.cpu().numpy(), site2.cpu().numpy())
                    assert np.array_equal(mod.cpu().numpy(), mod2.cpu().numpy())
    finally:
        eng.set_scan_driver(0)
        eng.set_table_variant(0)


def test_launch_timing_modes(engines):
    """m6a_profile_enable: HIP events around the encoder launches, the pooling launches, or both (bench.py's live
    roofline takes the encoder's inside the timed region and the pooling kernel's from extra steps)."""
    import torch
    eng = engines["hct116"]
    dev = torch.device("cuda:0")
    d = synthetic.make_sites(20000, 20, seed=3)
    tX, tk, to = (torch.from_numpy(d[k]).to(dev) for k in ("X", "site_kmers", "off"))
    for mode, want in ((True, (3, 3)), ("encoder", (3, 0)), ("pooling", (0, 3)), (False, (0, 0))):
        eng.profile(mode)
        for _ in range(3):
            eng.infer(tX, tk, to, 100)
        enc_ms, enc_n = eng.profile_read(0)
        pool_ms, pool_n = eng.profile_read(1)
        assert (enc_n, pool_n) == want
        assert (enc_ms > 0) == (enc_n > 0) and (pool_ms > 0) == (pool_n > 0)
    eng.profile(False)


def test_host_offsets_hint_same_results_and_no_stream_sync(engines):
    """m6a_set_host_offsets: with the loader's host copy of off[] a device-pointer call takes the bag statistics from
    it (checked on the device) instead of reading them back.  Results are bit-identical either way, for uniform and
    ragged bags and for the separate encode / pool entry points; the hint is consumed by one call."""
    import torch
    eng = engines["hek293t_glori"]
    dev = torch.device("cuda:0")
    for bags in (20, (20, 90), (1, 40)):
        d = synthetic.make_sites(3000, bags, seed=5)
        tX, tk, to = (torch.from_numpy(d[k]).to(dev) for k in ("X", "site_kmers", "off"))
        rp0, site0, mod0 = eng.infer(tX, tk, to, 100)
        eng.sync()
        want = [t.cpu().numpy() for t in (rp0, site0, mod0)]
        for _ in range(3):                                          # back to back, no sync in between
            eng.set_host_offsets(d["off"])
            rp1, site1, mod1 = eng.infer(tX, tk, to, 100)
        eng.sync()
        for a, b in zip(want, (rp1, site1, mod1)):
            assert np.array_equal(a, b.cpu().numpy(), equal_nan=True)
        eng.set_host_offsets(d["off"])
        rp2 = eng.get_read_probability(tX, tk, to)
        eng.set_host_offsets(d["off"])
        site2, mod2 = eng.calculate_site_proba(rp2, to, 100)
        eng.sync()
        assert np.array_equal(want[0], rp2.cpu().numpy()) and np.array_equal(want[1], site2.cpu().numpy(), equal_nan=True)
        assert np.array_equal(want[2], mod2.cpu().numpy(), equal_nan=True)


def test_host_offsets_hint_mismatch_is_reported(engines):
    """A host copy that is not the device array: the call's results are undefined, the next sync says so, and the
    context works normally afterwards."""
    import torch
    from m6anet_amd._lib import M6AError
    eng = engines["hek293t_glori"]
    dev = torch.device("cuda:0")
    d = synthetic.make_sites(2000, (20, 60), seed=9)
    tX, tk, to = (torch.from_numpy(d[k]).to(dev) for k in ("X", "site_kmers", "off"))
    rp0, site0, mod0 = eng.infer(tX, tk, to, 50)
    eng.sync()
    wrong = d["off"].copy()
    wrong[1000] += 1                                                # same range, same total, another histogram
    assert wrong[1000] <= wrong[1001]
    eng.set_host_offsets(wrong)
    eng.infer(tX, tk, to, 50)
    with pytest.raises(M6AError, match="m6a_set_host_offsets"):
        eng.sync()
    rp1, site1, mod1 = eng.infer(tX, tk, to, 50)                    # no hint: read back as before
    eng.sync()
    assert np.array_equal(site0.cpu().numpy(), site1.cpu().numpy()) and np.array_equal(mod0.cpu().numpy(), mod1.cpu().numpy())
    eng.set_host_offsets(np.arange(1, len(d["off"]) + 1, dtype=np.int64))   # off[0] != 0: rejected by the call itself
    with pytest.raises(M6AError, match="off"):
        eng.infer(tX, tk, to, 50)
    eng.sync()


# ------------------------------------------------------------------ full size ------------------
def test_full_size_properties(eng, orc, weights):
    """BASELINE.json configs[2] size (1M sites x 20 reads, T=1000): size-independent checks.
    Flush groups are independent, so the oracle can check randomly chosen groups exactly."""
    import torch
    S, T = 1_000_000, 1000
    d = synthetic.make_sites(S, 20, seed=20250328)
    dev = torch.device("cuda:0")
    X = torch.from_numpy(d["X"]).to(dev)
    km = torch.from_numpy(d["site_kmers"]).to(dev)
    off = torch.from_numpy(d["off"]).to(dev)
    eng.use_torch_stream()
    try:
        rp, site, mod = eng.infer(X, km, off, T)
        eng.sync()
        rp2, site2, mod2 = eng.infer(X, km, off, T)
        eng.sync()
    finally:
        eng.set_stream(None)
    rp, site, mod = rp.cpu().numpy(), site.cpu().numpy(), mod.cpu().numpy()
    # determinism
    assert np.array_equal(site, site2.cpu().numpy()) and np.array_equal(rp, rp2.cpu().numpy())
    assert np.all(np.isfinite(site)) and site.min() >= 0 and site.max() <= 1
    # exact oracle check on a sample of flush groups (group g>=1 = sites 16+32(g-1) .. +32)
    g = np.random.Generator(np.random.PCG64(1))
    for grp in [0, 1, 2] + list(g.integers(3, (S - 16) // 32, size=40)):
        a, b = (0, 16) if grp == 0 else (16 + 32 * (grp - 1), 16 + 32 * grp)
        sl = slice(a * 20, b * 20)
        p = orc.encode_reads(weights["hct116"], d["X"][sl], d["site_kmers"][a:b], d["off"][a:b + 1] - a * 20)
        assert np.allclose(rp[sl], p, rtol=1e-5, atol=1e-8)
        # a group replayed alone (batch_size = its size, so it is one flush group) sees the same stream
        w_site, w_mod = orc.site_pool(rp[sl], d["off"][a:b + 1] - a * 20, T, THR, batch_size=b - a, save_per_batch=2)
        assert same_sites(site[a:b], w_site)
        assert np.array_equal(mod[a:b], w_mod)
    # Monte-Carlo mean vs the closed form E = 1 - mean_i(1-p_i)... per site: 1 - (mean(1-p))^20
    q = (1.0 - rp.astype(np.float64)).reshape(S, 20).mean(axis=1)
    closed = 1.0 - q ** 20
    assert np.abs(site - closed).mean() < 0.01
    # last site is written (the reference would drop the final batch when #batches is even)
    assert site[-1] > 0


def test_full_size_ragged_properties(engines, orc, weights):
    """BASELINE.json configs[4], per-GPU shape (125k sites x 50..500 reads, HEK293T weights, T=1000):
    the index-table kernel against the stream-replaying scan kernel on EVERY site (bit for bit), determinism,
    and the oracle on sampled flush groups."""
    import torch
    eng = engines["hek293t_glori"]
    S, T = 125_000, 1000
    d = synthetic.make_sites(S, (50, 500), seed=20250328)
    dev = torch.device("cuda:0")
    X = torch.from_numpy(d["X"]).to(dev)
    km = torch.from_numpy(d["site_kmers"]).to(dev)
    off = torch.from_numpy(d["off"]).to(dev)
    eng.use_torch_stream()
    try:
        rp, site, mod = eng.infer(X, km, off, T)
        eng.sync()
        assert eng.last_pool_variant == "ragged-table"
        rp2, site2, mod2 = eng.infer(X, km, off, T)
        eng.sync()
        eng.set_scan_driver(1)
        site3, mod3 = eng.calculate_site_proba(rp, off, T)
        eng.sync()
        assert eng.last_pool_variant == "scan-group"
    finally:
        eng.set_scan_driver(0)
        eng.set_stream(None)
    rp, site, mod = rp.cpu().numpy(), site.cpu().numpy(), mod.cpu().numpy()
    assert np.array_equal(site, site2.cpu().numpy()) and np.array_equal(mod, mod2.cpu().numpy())
    assert np.array_equal(site, site3.cpu().numpy()) and np.array_equal(mod, mod3.cpu().numpy())
    assert np.all(np.isfinite(site)) and site.min() >= 0 and site.max() <= 1
    g = np.random.Generator(np.random.PCG64(2))
    o = d["off"]
    for grp in [0, 1, (S - 16) // 32] + list(g.integers(2, (S - 16) // 32, size=12)):
        a, b = (0, 16) if grp == 0 else (16 + 32 * (grp - 1), min(S, 16 + 32 * grp))
        sl = slice(int(o[a]), int(o[b]))
        p = orc.encode_reads(weights["hek293t_glori"], d["X"][sl], d["site_kmers"][a:b], o[a:b + 1] - o[a])
        assert np.allclose(rp[sl], p, rtol=1e-5, atol=1e-8)
        w_site, w_mod = orc.site_pool(rp[sl], o[a:b + 1] - o[a], T, THR, batch_size=b - a, save_per_batch=2)
        assert same_sites(site[a:b], w_site)
        assert np.array_equal(mod[a:b], w_mod)
    assert site[-1] > 0


@pytest.mark.parametrize("config,n_sites,bag,model", [("configs[3]", 8_000_000, 20, "hct116"),
                                                       ("configs[4]", 1_000_000, (50, 500), "hek293t_glori")])
def test_whole_8gpu_job_as_eight_shards_on_one_gpu(engines, orc, weights, config, n_sites, bag, model):
    """BASELINE.json configs[3] (8 M sites x 20 reads) and configs[4] (1 M sites x 50..500 reads, HEK293T weights) at
    FULL size, T = 1000, on the one GPU a test box has (5.8 / 9.9 GB of features, generated on the device): the job run
    whole, and run as the eight flush-group-aligned shards m6a_shard_plan gives the eight ranks, each with its job
    offset -- everything a rank computes on an 8-GPU node, minus the RCCL transport.  Shards = whole bit for bit, and
    sampled flush groups against the oracle."""
    import torch
    from m6anet_amd.engine import shard_plan
    eng = engines[model]
    dev = torch.device("cuda:0")
    T = 1000
    n_reads = synthetic.bag_sizes(n_sites, bag)
    off_h = np.zeros(n_sites + 1, np.int64)
    np.cumsum(n_reads, out=off_h[1:])
    R = int(off_h[-1])
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    X = torch.empty((R, 9), dtype=torch.float32, device=dev)
    for a in range(0, R, 1 << 24):                                    # in pieces: normal_ on 2.5 G elements at once is slow
        X[a:a + (1 << 24)].normal_(generator=gen)
    X.clamp_(-6.0, 6.0)
    km_h = synthetic.make_sites(n_sites, 1, seed=3)["site_kmers"]     # only the k-mer ids of this call are used
    km, off = torch.from_numpy(km_h).to(dev), torch.from_numpy(off_h).to(dev)
    eng.use_torch_stream()
    try:
        rp, site, mod = eng.infer(X, km, off, T)
        eng.sync()
        cuts = shard_plan(off_h, 8)
        assert cuts[0] == 0 and cuts[-1] == n_sites and np.all(np.diff(cuts) > 0)
        reads = np.diff(off_h[cuts])
        assert reads.max() / reads.mean() < 1.01                       # balanced by reads
        for r in range(8):
            a, b = int(cuts[r]), int(cuts[r + 1])
            eng.set_job_offset(a)
            o = off[a:b + 1] - off[a]
            rp_s, site_s, mod_s = eng.infer(X[off_h[a]:off_h[b]], km[a:b], o.contiguous(), T)
            eng.sync()
            assert torch.equal(site_s, site[a:b]) and torch.equal(mod_s, mod[a:b]), (config, r)
            assert torch.equal(rp_s, rp[off_h[a]:off_h[b]]), (config, r)
    finally:
        eng.set_job_offset(0)
        eng.set_stream(None)
    site_h, mod_h = site.cpu().numpy(), mod.cpu().numpy()
    assert np.all(np.isfinite(site_h)) and site_h.min() >= 0 and site_h.max() <= 1
    g = np.random.Generator(np.random.PCG64(4))
    n_groups = (n_sites - 16) // 32
    for grp in [0, n_groups] + list(g.integers(1, n_groups, size=6)):
        a, b = (0, 16) if grp == 0 else (16 + 32 * (grp - 1), min(n_sites, 16 + 32 * grp))
        lo, hi = int(off_h[a]), int(off_h[b])
        Xs, rps = X[lo:hi].cpu().numpy(), rp[lo:hi].cpu().numpy()
        p = orc.encode_reads(weights[model], Xs, km_h[a:b], off_h[a:b + 1] - lo)
        assert np.allclose(rps, p, rtol=1e-5, atol=1e-8)
        w_site, w_mod = orc.site_pool(rps, off_h[a:b + 1] - lo, T, THR, batch_size=b - a, save_per_batch=2)
        assert same_sites(site_h[a:b], w_site) and np.array_equal(mod_h[a:b], w_mod)


def test_beyond_4GiB_of_features(eng, orc, weights):
    """A job whose feature array crosses 2^32 bytes (3.4 M sites x 37 reads = 126 M reads, 4.5 GB of X): every
    byte offset on the path must be 64-bit (the 12-slot encoder and the register pooling kernel keep 32-bit tile /
    site indices and lane offsets only).  Random flush groups, including the last ones, against the oracle."""
    import torch
    S, n, T = 3_400_000, 37, 6
    g = np.random.Generator(np.random.PCG64(5))
    X = g.standard_normal((S * n, 9), dtype=np.float32)
    np.clip(X, -6, 6, out=X)
    km = g.integers(0, 66, size=(S, 3)).astype(np.uint8)
    off = np.arange(S + 1, dtype=np.int64) * n
    assert X.nbytes > 2 ** 32
    dev = torch.device("cuda:0")
    tX, tk, to = torch.from_numpy(X).to(dev), torch.from_numpy(km).to(dev), torch.from_numpy(off).to(dev)
    eng.use_torch_stream()
    try:
        rp, site, mod = eng.infer(tX, tk, to, T)
        eng.sync()
    finally:
        eng.set_stream(None)
    assert eng.last_encoder_kernel == "enc_site16_kernel" and eng.last_pool_variant == "ragged-table"   # n = 37 > 32: no uniform-bag kernel
    rp, site, mod = rp.cpu().numpy(), site.cpu().numpy(), mod.cpu().numpy()
    del tX
    n_groups = (S - 16) // 32
    for grp in [0, 1, n_groups - 1, n_groups] + list(g.integers(2, n_groups - 1, size=12)):
        a = 0 if grp == 0 else 16 + 32 * (grp - 1)
        b = min(S, 16 if grp == 0 else 16 + 32 * grp)
        sl = slice(a * n, b * n)
        p = orc.encode_reads(weights["hct116"], X[sl], km[a:b], off[a:b + 1] - a * n)
        assert np.array_equal(rp[sl].view(np.uint32), p.view(np.uint32)), grp           # auto = the oracle's bits, beyond 4 GiB of features too
        w_site, w_mod = orc.site_pool(rp[sl], off[a:b + 1] - a * n, T, THR, batch_size=b - a, save_per_batch=2)
        assert same_sites(site[a:b], w_site), grp
        assert np.array_equal(mod[a:b], w_mod), grp


# ------------------------------------------------------------------ CLI, config #1 --------------
def _run_cli(tmp_path, extra):
    import os
    from m6anet_amd.__main__ import main
    out = str(tmp_path / "out")
    data = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_tests_data")
    main(["inference", "--input_dir", data, "--out_dir", out, "--n_processes", "1"] + extra)
    return out


def test_cli_config1_matches_reference_csvs(tmp_path, golden):
    """BASELINE.json configs[0]: bundled data -> inference at num_iterations=5 -- here on the GPU --
    against the exact CSVs the reference wrote (tests/golden/config1_*.csv)."""
    import gzip
    import os
    import pandas as pd
    out = _run_cli(tmp_path, ["--num_iterations", "5"])
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    site = pd.read_csv(os.path.join(out, "data.site_proba.csv"))
    ref_site = pd.read_csv(os.path.join(gold, "config1_site_proba.csv"))
    assert list(site.columns) == list(ref_site.columns) and len(site) == len(ref_site) == 101
    for col in ("transcript_id", "transcript_position", "n_reads", "kmer"):      # bit-exact ids / k-mers
        assert (site[col] == ref_site[col]).all(), col
    assert np.array_equal(site["mod_ratio"], ref_site["mod_ratio"])
    assert np.abs(site["probability_modified"] - ref_site["probability_modified"]).max() <= 1e-5
    ours = open(os.path.join(out, "data.indiv_proba.csv")).read().splitlines()
    ref = gzip.open(os.path.join(gold, "config1_indiv_proba.csv.gz"), "rt").read().splitlines()
    assert len(ours) == len(ref) and ours[0] == ref[0]
    pa = np.array([float(r.rsplit(",", 1)[1]) for r in ours[1:]])
    pb = np.array([float(r.rsplit(",", 1)[1]) for r in ref[1:]])
    assert [r.rsplit(",", 1)[0] for r in ours[1:]] == [r.rsplit(",", 1)[0] for r in ref[1:]]   # read_index exact
    assert np.allclose(pa, pb, rtol=1e-5, atol=1e-8)
    # the CLI's default encoder performs the reference's float32 operations in the reference's order to the last one: the
    # site CSV is the reference's BYTE FOR BYTE, and so is every line of the per-read CSV but those of the few reads MKL's
    # thread partition computed outside its groups of four (15 of 5 595 in the capture)
    assert open(os.path.join(out, "data.site_proba.csv")).read().splitlines() == open(os.path.join(gold, "config1_site_proba.csv")).read().splitlines()
    assert sum(a != b for a, b in zip(ours, ref)) <= 30


def test_cli_reference_test_suite_bar(tmp_path):
    """The reference's own integration test (m6anet/tests/test_inference.py:10-37) re-stated on its
    own golden files: ids exact, read probabilities allclose, mod_ratio allclose, site probabilities
    atol=1e-2 at num_iterations=10000."""
    import os
    import pandas as pd
    out = _run_cli(tmp_path, ["--num_iterations", "10000"])
    data = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_tests_data")
    key_i = ["transcript_id", "transcript_position", "read_index"]
    key_s = ["transcript_id", "transcript_position"]
    ti = pd.read_csv(os.path.join(out, "data.indiv_proba.csv")).sort_values(key_i).reset_index(drop=True)
    ts = pd.read_csv(os.path.join(out, "data.site_proba.csv")).sort_values(key_s).reset_index(drop=True)
    gi = pd.read_csv(os.path.join(data, "data.indiv_proba.csv.gz")).sort_values(key_i).reset_index(drop=True)
    gs = pd.read_csv(os.path.join(data, "data.site_proba.csv.gz")).sort_values(key_s).reset_index(drop=True)
    for k in key_i:
        assert np.all(gi[k] == ti[k])
    assert np.allclose(gi["probability_modified"], ti["probability_modified"])
    for k in key_s:
        assert np.all(gs[k] == ts[k])
    assert np.allclose(gs["mod_ratio"], ts["mod_ratio"])
    assert np.allclose(gs["probability_modified"], ts["probability_modified"], atol=1e-2)


def test_cli_replicates_match_reference_csvs(tmp_path):
    """Replicate inference on the GPU (m6anet/tests/test_inference.py:40-82; NanopolishReplicateDS,
    m6anet/utils/data_utils.py:293-427): two input directories -> pooled bags, `<read>_<replicate>` ids.
    Against the CSVs the reference wrote for the same two directories (tests/golden/replicate_*.csv, T=5,
    n_processes=1): ids and suffixes exact, read probabilities rtol 1e-5, site probabilities <= 1e-5,
    mod_ratio exact."""
    import gzip
    import os
    import shutil
    import pandas as pd
    from m6anet_amd.__main__ import main
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    data = os.path.join(gold, "ref_tests_data")
    rep = tmp_path / "rep1"
    rep.mkdir()
    for fn in ("data.info", "data.json"):
        shutil.copyfile(os.path.join(data, fn), rep / fn)
    out = str(tmp_path / "out")
    main(["inference", "--input_dir", data, str(rep), "--out_dir", out, "--n_processes", "1", "--num_iterations", "5"])
    site = pd.read_csv(os.path.join(out, "data.site_proba.csv"))
    ref_site = pd.read_csv(os.path.join(gold, "replicate_site_proba.csv"))
    assert list(site.columns) == list(ref_site.columns) and len(site) == len(ref_site)
    for col in ("transcript_id", "transcript_position", "n_reads", "kmer"):
        assert (site[col] == ref_site[col]).all(), col
    assert np.array_equal(site["mod_ratio"], ref_site["mod_ratio"])
    assert np.abs(site["probability_modified"] - ref_site["probability_modified"]).max() <= 1e-5
    ours = open(os.path.join(out, "data.indiv_proba.csv")).read().splitlines()
    ref = gzip.open(os.path.join(gold, "replicate_indiv_proba.csv.gz"), "rt").read().splitlines()
    assert len(ours) == len(ref) and ours[0] == ref[0]
    ids_ours = [r.rsplit(",", 1)[0] for r in ours[1:]]
    assert ids_ours == [r.rsplit(",", 1)[0] for r in ref[1:]]                      # `966210_0` / `966210_1` exact
    assert {i.rsplit("_", 1)[1] for i in ids_ours} == {"0", "1"}
    pa = np.array([float(r.rsplit(",", 1)[1]) for r in ours[1:]])
    pb = np.array([float(r.rsplit(",", 1)[1]) for r in ref[1:]])
    assert np.allclose(pa, pb, rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("T,bs,spb,seed", [(30, 51, 2, 0), (50, 8, 3, 0), (20, 13, 2, 7)])
def test_cli_drop_unflushed_tail_writes_the_reference_row_set(tmp_path, golden, T, bs, spb, seed):
    """--drop_unflushed_tail: exactly the rows the reference wrote for this geometry (its inverted flush test,
    m6anet/utils/inference_utils.py:47, loses the batches after the last flush -- batch 51 writes 51 of the 101
    sites), with the reference's values; without the flag every site is written."""
    import os
    import pandas as pd
    g = golden("bundled_site.npz")
    key = "T%d_bs%d_spb%d_seed%d" % (T, bs, spb, seed)
    written = g[key + "_written"].astype(bool)
    args = ["--num_iterations", str(T), "--batch_size", str(bs), "--save_per_batch", str(spb), "--seed", str(seed)]
    out = _run_cli(tmp_path, args + ["--drop_unflushed_tail"])
    site = pd.read_csv(os.path.join(out, "data.site_proba.csv"))
    n = int(written.sum())
    assert len(site) == n and written[:n].all() and not written[n:].any()
    assert np.abs(site["probability_modified"].values - g[key + "_site"][:n]).max() <= 1e-5
    assert np.abs(site["mod_ratio"].values - g[key + "_mod"][:n]).max() <= 2e-16      # the CSV holds '%.16f' text
    indiv = pd.read_csv(os.path.join(out, "data.indiv_proba.csv"))
    assert len(indiv) == int(site["n_reads"].sum())
    full = pd.read_csv(os.path.join(_run_cli(tmp_path / "all", args), "data.site_proba.csv"))
    assert len(full) == 101 and full.iloc[:n].equals(site)


def test_cli_from_binary_site_store_equals_cli_from_json(tmp_path):
    """pack -> inference from the mapped store: byte-identical CSVs to inference from data.json / data.info."""
    import os
    from m6anet_amd.__main__ import main
    data = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_tests_data")
    store = str(tmp_path / "bundled.m6astore")
    main(["pack", "--input_dir", data, "--out", store])
    a = _run_cli(tmp_path / "a", ["--num_iterations", "50"])
    b = str(tmp_path / "b" / "out")
    main(["inference", "--input_dir", store, "--out_dir", b, "--n_processes", "1", "--num_iterations", "50"])
    for fn in ("data.site_proba.csv", "data.indiv_proba.csv"):
        assert open(os.path.join(a, fn), "rb").read() == open(os.path.join(b, fn), "rb").read(), fn
    with pytest.raises(ValueError, match="re-run"):                      # a store holds features normalised for one model
        main(["inference", "--input_dir", store, "--out_dir", b, "--pretrained_model", "arabidopsis_RNA002"])


def test_cli_rejects_cpu_device(tmp_path):
    with pytest.raises(ValueError):
        _run_cli(tmp_path, ["--device", "cpu"])


def test_full_pipeline_from_eventalign(tmp_path):
    """eventalign.txt -> native dataprep -> inference CLI on the GPU, against the reference's golden
    CSVs.  Read order inside a site differs from the reference's data.json (machine-dependent there),
    so: ids / n_reads / k-mers exact, read probabilities equal as per-site multisets (allclose),
    mod_ratio exact, site probabilities at the reference's own bar (atol 1e-2, T=10000)."""
    import gzip
    import os
    import pandas as pd
    from m6anet_amd.__main__ import main
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_tests_data")
    ev = tmp_path / "eventalign.txt"
    ev.write_bytes(gzip.open(os.path.join(gold, "eventalign.txt.gz"), "rb").read())
    prep, out = str(tmp_path / "prep"), str(tmp_path / "out")
    main(["dataprep", "--eventalign", str(ev), "--out_dir", prep, "--n_processes", "0", "--min_segment_count", "1"])
    main(["inference", "--input_dir", prep, "--out_dir", out, "--num_iterations", "10000"])
    key_s = ["transcript_id", "transcript_position"]
    ts = pd.read_csv(os.path.join(out, "data.site_proba.csv")).sort_values(key_s).reset_index(drop=True)
    gs = pd.read_csv(os.path.join(gold, "data.site_proba.csv.gz")).sort_values(key_s).reset_index(drop=True)
    for k in key_s + ["n_reads", "kmer"]:
        assert (ts[k] == gs[k]).all(), k
    assert np.allclose(ts["mod_ratio"], gs["mod_ratio"])
    assert np.allclose(ts["probability_modified"], gs["probability_modified"], atol=1e-2)
    key_i = key_s + ["read_index"]
    ti = pd.read_csv(os.path.join(out, "data.indiv_proba.csv")).sort_values(key_i).reset_index(drop=True)
    gi = pd.read_csv(os.path.join(gold, "data.indiv_proba.csv.gz")).sort_values(key_i).reset_index(drop=True)
    assert (ti[key_i].values == gi[key_i].values).all()
    assert np.allclose(ti["probability_modified"], gi["probability_modified"], rtol=2e-5, atol=1e-7)
