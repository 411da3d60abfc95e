"""Domain constants of the m6A inference path.

Mirrors the values the reference keeps in m6anet/utils/constants.py:8-37 (pretrained
registry, thresholds, min reads, the 66-entry 5-mer vocabulary and the 18 DRACH motifs)
and m6anet/utils/data_utils.py:89-96 (same vocabulary built inside the dataset).
Nothing is imported from the reference; tests/test_abi_and_host.py checks the vocabulary
against tests/golden/vocab66.txt (captured from the reference).
"""
import os
from itertools import product

ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")

DEFAULT_MIN_READS = 20              # constants.py:14
N_SAMPLES = 20                      # hard-coded at inference_utils.py:54
DEFAULT_READ_THRESHOLD = 0.033379376            # constants.py:15
ARABIDOPSIS_READ_THRESHOLD = 0.0032978046219796  # constants.py:19
NUM_NEIGHBORING_FEATURES = 1

# DRACH: D=[AGT] R=[GA] A C H=[ACT]; one flanking base either side -> 7-mers
_D, _R, _H, _N = "AGT", "GA", "ACT", "ACGT"
M6A_KMERS = ["".join(x) for x in product(_D, _R, "A", "C", _H)]           # 18 centre 5-mers
ALL_7MERS = sorted("".join(x) for x in product(_N, _D, _R, "A", "C", _H, _N))  # 288
ALL_KMERS = sorted({s[i:i + 5] for s in ALL_7MERS for i in range(3)})        # 66-word vocab
KMER_TO_INT = {k: i for i, k in enumerate(ALL_KMERS)}
INT_TO_KMER = {i: k for i, k in enumerate(ALL_KMERS)}
N_VOCAB = len(ALL_KMERS)

# name -> (weights blob, read threshold, norm-factor file)   (constants.py:24-27)
PRETRAINED_CONFIGS = {
    "HCT116_RNA002": ("weights_hct116.bin", DEFAULT_READ_THRESHOLD, "norm_hct116.npz"),
    "arabidopsis_RNA002": ("weights_arabidopsis.bin", ARABIDOPSIS_READ_THRESHOLD, "norm_arabidopsis.npz"),
    "HEK293T_RNA004": ("weights_hek293t_glori.bin", DEFAULT_READ_THRESHOLD, "norm_hct116.npz"),
    "HEK293T_RNA004_M6ACE": ("weights_hek293t_m6ace.bin", DEFAULT_READ_THRESHOLD, "norm_hct116.npz"),
}
DEFAULT_PRETRAINED_MODEL = "HCT116_RNA002"
# the reference only accepts these three on the command line (constants.py:8, inference.py:77-78)
DEFAULT_PRETRAINED_MODELS = ["HCT116_RNA002", "arabidopsis_RNA002", "HEK293T_RNA004"]

N_WEIGHT_FLOATS = 7997   # E 66x2, W1 150x15, b1, gamma, beta, mu, var (150 each), W2 32x150, b2 32, W3 32, b3


def asset_path(name):
    return os.path.join(ASSET_DIR, name)


def kmer7_to_ids(kmer7):
    """7-mer -> the three overlapping 5-mer vocabulary ids (data_utils.py:195-196,223)."""
    return [KMER_TO_INT[kmer7[i:i + 5]] for i in range(3)]
