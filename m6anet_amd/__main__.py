"""`python -m m6anet_amd {dataprep,pack,inference} ...` -- the hot path and the steps before it
(dispatcher shape of m6anet/__init__.py:11-30)."""
import sys

from . import _early

if __name__ == "__main__":
    _early.maybe_start(sys.argv[1:])        # `inference --gpus N`: the other ranks start before the heavy imports below

from argparse import ArgumentParser  # noqa: E402

from .scripts import dataprep, inference, pack  # noqa: E402


def main(argv=None):
    parser = ArgumentParser(prog="m6anet_amd")
    sub = parser.add_subparsers(dest="command", required=True)
    sub.add_parser("inference", parents=[inference.argparser()], help="run the MI355X inference hot path")
    sub.add_parser("dataprep", parents=[dataprep.argparser()], help="eventalign.txt -> data.json / data.info (native, host-only)")
    sub.add_parser("pack", parents=[pack.argparser()], help="data.json / data.info -> one binary site store that later runs map")
    args = parser.parse_args(argv)
    {"inference": inference, "dataprep": dataprep, "pack": pack}[args.command].main(args)


if __name__ == "__main__":
    main(sys.argv[1:])
