"""`python -m m6anet_amd inference ...` -- the one sub-command on the hot path
(dispatcher shape of m6anet/__init__.py:11-30)."""
import sys
from argparse import ArgumentParser

from .scripts import inference


def main(argv=None):
    parser = ArgumentParser(prog="m6anet_amd")
    sub = parser.add_subparsers(dest="command", required=True)
    sub.add_parser("inference", parents=[inference.argparser()], help="run the MI355X inference hot path")
    args = parser.parse_args(argv)
    inference.main(args)


if __name__ == "__main__":
    main(sys.argv[1:])
