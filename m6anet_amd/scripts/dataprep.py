"""`dataprep` sub-command: same flags as `m6anet dataprep` (m6anet/scripts/dataprep.py:14-51),
running the native implementation in libm6a_io.so (host-only)."""
import os
from argparse import ArgumentDefaultsHelpFormatter, ArgumentParser

from ..constants import NUM_NEIGHBORING_FEATURES


def argparser():
    parser = ArgumentParser(formatter_class=ArgumentDefaultsHelpFormatter, add_help=False)
    parser.add_argument("--eventalign", required=True, help="eventalign filepath, the output from nanopolish.")
    parser.add_argument("--out_dir", required=True, help="output directory.")
    parser.add_argument("--n_processes", default=1, type=int, help="number of host threads (0 = all).")
    parser.add_argument("--chunk_size", default=1000000, type=int,
                        help="accepted for compatibility (the native indexer streams the file once).")
    parser.add_argument("--readcount_min", default=1, type=int, help="minimum read counts per gene.")
    parser.add_argument("--readcount_max", default=1000, type=int, help="maximum read counts per gene.")
    parser.add_argument("--min_segment_count", default=20, type=int,
                        help="minimum read counts over each candidate m6A segment.")
    parser.add_argument("--skip_index", default=False, action="store_true",
                        help="skip indexing the eventalign nanopolish output (reuse eventalign.index).")
    parser.add_argument("--n_neighbors", default=NUM_NEIGHBORING_FEATURES, type=int,
                        help="number of neighboring features to extract (1..16; the shipped models take 1).")
    parser.add_argument("--compress", default=False, action="store_true",
                        help="round down the features to 3 decimal places.")
    return parser


def main(args):
    from .. import _io
    if not os.path.exists(args.out_dir):
        os.makedirs(args.out_dir)
    _io.dataprep(args.eventalign, args.out_dir, n_threads=args.n_processes, readcount_min=args.readcount_min,
                 readcount_max=args.readcount_max, min_segment_count=args.min_segment_count,
                 n_neighbors=args.n_neighbors, compress=args.compress, skip_index=args.skip_index)
