"""Pins the CPU oracle (oracle/m6a_oracle.c) against vectors captured from the imported
reference (tests/golden/make_golden.py) and against NumPy's own legacy RandomState.

Tolerances:
  * RNG words / choice indices / flush groups: bit-exact.
  * read probabilities: np.allclose default rtol=1e-5, atol=1e-8 -- the reference's own bar
    (m6anet/tests/test_inference.py:32).
  * site probabilities given identical read probabilities: bit-exact (float32), because the
    oracle replays MT19937 + masked rejection + sequential float32 product + NumPy pairwise
    mean; with the oracle's own read probabilities: 1e-6 abs.
"""
import os

import numpy as np
import pytest

from oracle import m6a_oracle as orc
from m6anet_amd import synthetic
from m6anet_amd.constants import DEFAULT_READ_THRESHOLD

THR = np.float32(DEFAULT_READ_THRESHOLD)


def test_mt19937_raw_words(golden):
    g = golden("rng_known.npz")
    for seed in (0, 1, 42, 20250328, 4294967295):
        assert np.array_equal(orc.mt_raw(seed, 4096), g[f"raw_seed{seed}"])
    # seed 0 first words, as quoted in SURVEY.md section 0.3
    assert orc.mt_raw(0, 4).tolist() == [2357136044, 2546248239, 3071714933, 3626093760]


def test_choice_known_answers(golden):
    g = golden("rng_known.npz")
    for n in (1, 2, 3, 20, 23, 32, 33, 64, 65, 500, 662, 1000, 1024, 1025, 70000):
        (idx,) = orc.choice_stream(0, [(n, 4096)])
        assert np.array_equal(idx, g[f"choice_seed0_n{n}"]), n
    seq = orc.choice_stream(42, [(20, 100), (33, 100), (1, 100), (64, 100), (21, 100)])
    assert np.array_equal(np.concatenate(seq), g["choice_seed42_seq_20_33_1_64_21"])


@pytest.mark.parametrize("n", [2, 7, 20, 31, 32, 33, 100, 257, 999])
def test_choice_against_live_numpy(n):
    np.random.seed(123)
    want = np.random.choice(np.arange(n), 5000, replace=True)
    (got,) = orc.choice_stream(123, [(n, 5000)])
    assert np.array_equal(got, want)


@pytest.mark.parametrize("T", [1, 5, 7, 8, 9, 100, 127, 128, 129, 255, 1000, 1003, 4096, 10000])
def test_pairwise_mean_matches_numpy(T):
    a = np.random.Generator(np.random.PCG64(T)).random(T, dtype=np.float32)
    assert orc.pairwise_sum(a) == a.sum()
    assert np.float32(orc.pairwise_sum(a) / np.float32(T)) == a.mean()


def test_flush_groups():
    # default geometry: {0}, {1,2}, {3,4}, ... in batches of 16 sites
    assert orc.flush_groups(101, 16, 2).tolist() == [0, 16, 48, 80, 101]
    # even number of batches: the reference never writes the last batch; we close it as a group
    assert orc.flush_groups(101, 51, 2).tolist() == [0, 51, 101]
    assert orc.flush_groups(101, 13, 2).tolist() == [0, 13, 39, 65, 91, 101]
    assert orc.flush_groups(101, 8, 3).tolist() == [0, 8, 16, 32, 40, 56, 64, 80, 88, 101]
    # save_per_batch=1 never flushes in the reference: one group with everything
    assert orc.flush_groups(40, 16, 1).tolist() == [0, 40]
    assert orc.flush_groups(5, 16, 2).tolist() == [0, 5]


def test_read_probs_all_models(golden, weights):
    b = golden("bundled_inputs.npz")
    want = golden("bundled_readprob.npz")
    for name, w in weights.items():
        got = orc.encode_reads(w, b["X"], b["site_kmers"], b["off"])
        assert np.allclose(got, want[name], rtol=1e-5, atol=1e-8), name
        assert np.array_equal(got, orc.encode_reads(w, b["X"], b["site_kmers"], b["off"], n_threads=3))


def test_read_probs_vs_reference_golden_csv(golden, weights):
    """The reference's own golden file (m6anet/tests/data/data.indiv_proba.csv.gz), same bar as
    m6anet/tests/test_inference.py:29-32."""
    import pandas as pd
    b = golden("bundled_inputs.npz")
    got = orc.encode_reads(weights["hct116"], b["X"], b["site_kmers"], b["off"])
    n = np.diff(b["off"])
    df = pd.DataFrame({"transcript_id": np.repeat(b["tx_ids"], n), "transcript_position": np.repeat(b["tx_pos"], n),
                       "read_index": b["read_ids"].astype(np.int64), "p": got})
    ref = pd.read_csv(os.path.join(os.path.dirname(__file__), "golden", "ref_tests_data", "data.indiv_proba.csv.gz"))
    key = ["transcript_id", "transcript_position", "read_index"]
    df = df.sort_values(key).reset_index(drop=True)
    ref = ref.sort_values(key).reset_index(drop=True)
    assert (df[key].values == ref[key].values).all()
    assert np.allclose(ref["probability_modified"], df["p"])


CASES = [(5, 16, 2, 0), (100, 16, 2, 0), (1000, 16, 2, 0), (50, 8, 3, 0), (20, 13, 2, 7), (30, 51, 2, 0)]


@pytest.mark.parametrize("T,bs,spb,seed", CASES)
def test_site_probs_bit_exact_given_reference_read_probs(golden, T, bs, spb, seed):
    b = golden("bundled_inputs.npz")
    p = golden("bundled_readprob.npz")["hct116"]
    g = golden("bundled_site.npz")
    key = f"T{T}_bs{bs}_spb{spb}_seed{seed}"
    site, mod = orc.site_pool(p, b["off"], T, THR, seed, bs, spb)
    assert np.array_equal(site, g[key + "_site"])
    assert np.array_equal(mod, g[key + "_mod"])
    site4, _ = orc.site_pool(p, b["off"], T, THR, seed, bs, spb, n_threads=4)
    assert np.array_equal(site, site4)


def test_end_to_end_oracle_vs_reference_run(golden, weights):
    b = golden("bundled_inputs.npz")
    g = golden("bundled_site.npz")
    p = orc.encode_reads(weights["hct116"], b["X"], b["site_kmers"], b["off"])
    site, mod = orc.site_pool(p, b["off"], 1000, THR)
    assert np.abs(site - g["T1000_bs16_spb2_seed0_site"]).max() < 1e-6
    assert np.array_equal(mod, g["T1000_bs16_spb2_seed0_mod"])


def test_reference_own_site_golden(golden, weights):
    """m6anet/tests/data/data.site_proba.csv.gz at the reference's own tolerances
    (m6anet/tests/test_inference.py:34-37: mod_ratio allclose, site probability atol=1e-2)."""
    import pandas as pd
    b = golden("bundled_inputs.npz")
    p = orc.encode_reads(weights["hct116"], b["X"], b["site_kmers"], b["off"])
    site, mod = orc.site_pool(p, b["off"], 10000, THR)
    df = pd.DataFrame({"transcript_id": b["tx_ids"], "transcript_position": b["tx_pos"], "site": site, "mod": mod})
    ref = pd.read_csv(os.path.join(os.path.dirname(__file__), "golden", "ref_tests_data", "data.site_proba.csv.gz"))
    key = ["transcript_id", "transcript_position"]
    m = ref.merge(df, on=key)
    assert len(m) == len(ref) == 101
    assert np.allclose(m["mod_ratio"], m["mod"])
    assert np.allclose(m["probability_modified"], m["site"], atol=1e-2)


@pytest.mark.parametrize("tag,model,kw", [("uniform20", "hct116", dict(n_sites=1000, bag=20)),
                                          ("ragged", "hek293t_glori", dict(n_sites=200, bag=(50, 500)))])
def test_synthetic_small(golden, weights, tag, model, kw):
    g = golden("synthetic_small.npz")
    d = synthetic.make_sites(seed=20250328, **kw)
    assert np.array_equal(d["off"], g[f"{tag}_off"])
    assert g[f"{tag}_xsum"][0] == np.float64(d["X"].astype(np.float64).sum())   # same inputs as captured
    p = orc.encode_reads(weights[model], d["X"], d["site_kmers"], d["off"])
    assert np.allclose(p, g[f"{tag}_readprob"], rtol=1e-5, atol=1e-8)
    for T in (100, 1000):
        site, mod = orc.site_pool(g[f"{tag}_readprob"], d["off"], T, THR)
        assert np.array_equal(site, g[f"{tag}_site_T{T}"])
        assert np.array_equal(mod, g[f"{tag}_mod"])


def test_bag_forward(golden, weights):
    g = golden("bag_forward.npz")
    B = g["X"].shape[0]
    off = np.arange(B + 1, dtype=np.int64) * 20
    p = orc.encode_reads(weights["hct116"], g["X"].reshape(-1, 9), g["kmer"], off)
    assert np.allclose(orc.bag_noisy_or(p, 20), g["site_prob"], rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------ validation-style forward -----
VAL_CASES = [("seed0_T5_bs16", 0, 5, True), ("seed7_T3_bs101", 7, 3, True), ("seed1_T12_bs1", 1, 12, False)]


@pytest.mark.parametrize("key,seed,T,same_batching", VAL_CASES)
def test_validation_sampler_and_predictions_vs_reference_validate(golden, key, seed, T, same_batching):
    """`validate` (training_utils.py:213-268) at num_workers=0 -- the sampler without
    replacement (data_utils.py:213-214) index for index, the per-pass predictions and their mean."""
    g = golden("validate.npz")
    b = golden("bundled_inputs.npz")
    p = golden("bundled_readprob.npz")["hct116"]
    assert np.array_equal(np.diff(b["off"]), g["n_reads"])
    assert np.array_equal(orc.validation_indices(seed, b["off"], T), g[key + "_idx"])
    y, avg = orc.validate(p, b["off"], T, seed)
    if same_batching:
        # the reference encoded the sampled bags batch by batch; for these batch sizes its read probabilities
        # are bit-identical to the whole-job ones the fixture holds
        assert np.array_equal(y, g[key + "_y_pred"])
        assert np.array_equal(avg, g[key + "_y_pred_avg"])
    else:
        # batch_size 1: torch's sgemm on 20-row batches rounds a few read probabilities differently (1 ulp)
        assert np.abs(y - g[key + "_y_pred"]).max() <= 2.4e-7
        assert np.abs(avg - g[key + "_y_pred_avg"]).max() <= 2.4e-7
    # the mean is float32, pass after pass (np.mean over axis 0 of a C-contiguous array)
    acc = np.zeros(y.shape[1], np.float32)
    for row in g[key + "_y_pred"]:
        acc = acc + row
    assert np.array_equal(acc / np.float32(T), g[key + "_y_pred_avg"])


def test_validation_sampler_rejects_short_bags():
    with pytest.raises(ValueError):
        orc.validation_indices(0, np.array([0, 25, 44], np.int64), 2)


@pytest.mark.parametrize("key,seed,T,same_batching", VAL_CASES)
def test_validation_metrics_mirror(golden, key, seed, T, same_batching):
    """m6anet_amd.training_utils' roc/pr AUC and BCE against the reference's (sklearn / torch) numbers."""
    from m6anet_amd import training_utils as tu
    g = golden("validate.npz")
    avg, y = g[key + "_y_pred_avg"], g["y_true"]
    assert abs(tu.get_roc_auc(y, avg) - float(g[key + "_roc_auc"])) < 1e-12
    assert abs(tu.get_pr_auc(y, avg) - float(g[key + "_pr_auc"])) < 1e-12
    assert abs(tu.binary_cross_entropy(avg, y) - float(g[key + "_avg_loss"])) < 1e-6
