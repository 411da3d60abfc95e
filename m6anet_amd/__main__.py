"""`python -m m6anet_amd {dataprep,inference} ...` -- the hot path and the step before it
(dispatcher shape of m6anet/__init__.py:11-30)."""
import sys
from argparse import ArgumentParser

from .scripts import dataprep, inference


def main(argv=None):
    parser = ArgumentParser(prog="m6anet_amd")
    sub = parser.add_subparsers(dest="command", required=True)
    sub.add_parser("inference", parents=[inference.argparser()], help="run the MI355X inference hot path")
    sub.add_parser("dataprep", parents=[dataprep.argparser()], help="eventalign.txt -> data.json / data.info (native, host-only)")
    args = parser.parse_args(argv)
    {"inference": inference, "dataprep": dataprep}[args.command].main(args)


if __name__ == "__main__":
    main(sys.argv[1:])
