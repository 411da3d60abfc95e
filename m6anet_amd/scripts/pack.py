"""`pack` sub-command: data.info + data.json -> one binary site store (include/m6a_io.h, m6a_io_save_store).

The reference re-parses data.json on every run, one seek + json.loads + normalise per site
(m6anet/utils/data_utils.py:152-231: 0.52 ms per site); `m6anet_amd inference --input_dir <file>.m6astore` maps
the packed file instead.  The store holds features normalised for ONE model (its norm factors) and the sites that
passed the read-count filter, and says so in its header.
"""
import os
from argparse import ArgumentDefaultsHelpFormatter, ArgumentParser

from ..constants import DEFAULT_MIN_READS, DEFAULT_PRETRAINED_MODEL, DEFAULT_PRETRAINED_MODELS, PRETRAINED_CONFIGS
from ..data_utils import STORE_SUFFIX, pack_sites


def argparser():
    parser = ArgumentParser(formatter_class=ArgumentDefaultsHelpFormatter, add_help=False)
    parser.add_argument("--input_dir", nargs="+", required=True, help="directories containing data.info and data.json (several = replicates).")
    parser.add_argument("--out", default=None, help="store file to write (default: <first input_dir>/data%s)." % STORE_SUFFIX)
    parser.add_argument("--pretrained_model", default=DEFAULT_PRETRAINED_MODEL, type=str,
                        help="model whose normalisation factors are applied. Options include {}.".format(DEFAULT_PRETRAINED_MODELS))
    parser.add_argument("--norm_path", default=None, help="normalisation factors file; overrides --pretrained_model's.")
    parser.add_argument("--n_processes", default=0, type=int, help="loader threads (0 = all hardware threads).")
    return parser


def main(args):
    if args.pretrained_model not in PRETRAINED_CONFIGS:
        raise ValueError("Invalid pretrained model {}, must be one of {}".format(args.pretrained_model, DEFAULT_PRETRAINED_MODELS))
    norm = args.norm_path or PRETRAINED_CONFIGS[args.pretrained_model][2]
    out = args.out or os.path.join(args.input_dir[0], "data" + STORE_SUFFIX)
    nat = pack_sites(args.input_dir, out, DEFAULT_MIN_READS, norm, n_threads=args.n_processes)
    print("%s: %d sites, %d reads, %d replicate(s), %.1f MB" % (out, nat.tx_pos.size, nat.X.shape[0], nat.n_replicates,
                                                              os.path.getsize(out) / 1e6))
