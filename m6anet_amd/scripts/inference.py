"""`inference` sub-command: same flags, defaults and output files as `m6anet inference`
(m6anet/scripts/inference.py:20-106), running the hot path on an MI355X through libm6a_hip.so."""
import os
import pathlib
import warnings
from argparse import ArgumentDefaultsHelpFormatter, ArgumentParser

from ..constants import (DEFAULT_MIN_READS, DEFAULT_PRETRAINED_MODEL, DEFAULT_PRETRAINED_MODELS,
                         DEFAULT_READ_THRESHOLD, N_WEIGHT_FLOATS, PRETRAINED_CONFIGS)
from ..data_utils import STORE_SUFFIX, load_sites, load_sites_native, open_store
from ..engine import M6ANetEngine, load_weights, weights_from_state_dict
from ..inference_utils import INDIV_HEADER, SITE_HEADER, run_inference


def argparser():
    parser = ArgumentParser(formatter_class=ArgumentDefaultsHelpFormatter, add_help=False)
    parser.add_argument("--input_dir", nargs="*", required=True,
                        help="directories containing data.info and data.json, or ONE binary site store written by "
                             "`m6anet_amd pack` (*%s)." % STORE_SUFFIX)
    parser.add_argument("--out_dir", required=True, help="directory to output inference results.")
    parser.add_argument("--pretrained_model", default=DEFAULT_PRETRAINED_MODEL, type=str,
                        help="pre-trained model available at m6anet. Options include {}.".format(DEFAULT_PRETRAINED_MODELS))
    parser.add_argument("--model_config", default=None,
                        help="accepted for compatibility; only the m6anet.toml topology is supported.")
    parser.add_argument("--model_state_dict", default=None,
                        help="path to model weights: a reference .pt checkpoint (needs torch) or a flat .bin blob.")
    parser.add_argument("--norm_path", default=PRETRAINED_CONFIGS[DEFAULT_PRETRAINED_MODEL][2],
                        help="path to normalization factors file (.npz of this package or the reference's .joblib)")
    parser.add_argument("--batch_size", default=16, type=int, help="batch size for inference.")
    parser.add_argument("--save_per_batch", default=2, type=int,
                        help="saving inference results every save_per_batch multiples.")
    parser.add_argument("--n_processes", default=25, type=int,
                        help="accepted for compatibility; results are those of the reference at n_processes=1.")
    parser.add_argument("--num_iterations", default=1000, type=int, help="number of sampling run.")
    parser.add_argument("--device", default="cuda:0", type=str,
                        help="GPU to run on (cuda:N / hip:N). There is no CPU path.")
    parser.add_argument("--seed", default=0, type=int, help="random seed for sampling.")
    parser.add_argument("--read_proba_threshold", default=DEFAULT_READ_THRESHOLD, type=float,
                        help="default probability threshold for a read to be considered modified.")
    parser.add_argument("--gpus", default=1, type=int,
                        help="GPUs of this node to split the job's sites over: one process per GPU, flush-group-aligned shards, "
                             "every rank writes the rows of its own sites; the output does not depend on it.")
    parser.add_argument("--encoder", default=None, choices=["reference", "fast"],
                        help="read encoder kernel.  reference (the default, and the library's own default): the 16-slot kernels, which "
                             "perform the reference's float32 operations in the reference's order all the way to the sigmoid -- on the "
                             "host configuration it was validated against (torch + MKL on AVX-512) read probabilities are bit-identical "
                             "to `m6anet inference` wherever MKL groups a batch's rows in fours (every read of 20-read bags, > 99.9 %% "
                             "of ragged ones); activations beyond 2^64 saturate and a -inf pre-activation becomes NaN (DESIGN.md).  "
                             "fast: opt-in, for jobs whose bags all have >= 16 reads a 12-slot kernel 4-5 %% faster per step and within "
                             "rtol 1e-5 of the reference on every fixture (random fuzz: a few reads in 10^10 beyond that bar, worst 1.08 x).  "
                             "The encoder is under 1 %% of this command's wall time either way.  Without this flag the environment "
                             "variable M6A_ENCODER (auto|reference|general16|csite12|walk16|fast), if set, decides; an unknown value is an error.")
    parser.add_argument("--drop_unflushed_tail", action="store_true",
                        help="reference-compatible output: omit the batches after the reference's last flush, which "
                             "`m6anet inference` never writes (its flush test is inverted); default: write every site.")
    return parser


ENCODER_MODES = {"reference": 1, "fast": 4}           # m6a_set_encoder_variant (include/m6a.h)
ENCODER_ENV_VALUES = ("", "auto", "reference", "general16", "csite12", "walk16", "fast")


def make_engine_for(args, weights, device):
    """The context of one `inference` process with --encoder applied to IT (m6a_set_encoder_variant), not to the process's
    environment: a library caller's other contexts keep their own choice.  An explicit --encoder always wins; without the
    flag M6A_ENCODER decides if set (m6a_create reads it and refuses values it does not know -- checked here first, so the
    message names the variable), else `reference`, which is also what the library does when nobody says anything."""
    env = os.environ.get("M6A_ENCODER")
    if env is not None and env not in ENCODER_ENV_VALUES:
        raise ValueError("M6A_ENCODER=%s: must be one of %s" % (env, "|".join(v for v in ENCODER_ENV_VALUES if v)))
    engine = M6ANetEngine(weights=weights, device=device)
    choice = getattr(args, "encoder", None)
    if choice is not None:
        engine.set_encoder_variant(ENCODER_MODES[choice])
    elif env is None:
        engine.set_encoder_variant(ENCODER_MODES["reference"])
    return engine


def _device_index(device):
    d = str(device).lower()
    if d.startswith("cpu"):
        raise ValueError("--device cpu: this build runs the hot path on an MI355X only (no CPU fallback)")
    return int(d.split(":")[1]) if ":" in d else 0


def _check_model_config(path):
    """Only the m6anet.toml topology exists here (m6anet/model/configs/model_configs/m6anet.toml); any other
    block list would silently run the wrong network."""
    import tomli
    with open(path, "rb") as f:
        blocks = tomli.load(f).get("block", [])
    want = ["DeaggregateNanopolish", "KmerMultipleEmbedding", "ConcatenateFeatures", "Linear", "Linear", "SigmoidProdPooling"]
    got = [b.get("block_type") for b in blocks]
    dims = [(b.get("input_channel"), b.get("output_channel")) for b in blocks if b.get("block_type") == "Linear"]
    if got != want or dims != [(15, 150), (150, 32)]:
        raise ValueError("--model_config %s is not the m6anet.toml topology (blocks %s): not supported" % (path, got))


def main(args):
    if not 0 <= int(args.seed) <= 0xffffffff:
        raise ValueError("Seed must be between 0 and 2**32 - 1")        # what np.random.seed raises
    if args.model_config is not None:
        _check_model_config(args.model_config)
    if args.model_state_dict is not None:
        warnings.warn("--model_state_dict is specified, overwriting default model weights")
        if str(args.model_state_dict).endswith(".bin"):
            import numpy as np
            weights = np.fromfile(args.model_state_dict, np.float32)
            if weights.size != N_WEIGHT_FLOATS:
                raise ValueError("%s holds %d floats, expected %d (layout in include/m6a.h)"
                                 % (args.model_state_dict, weights.size, N_WEIGHT_FLOATS))
        else:
            import torch
            weights = weights_from_state_dict(torch.load(args.model_state_dict, map_location="cpu", weights_only=True))
    else:
        if args.pretrained_model not in DEFAULT_PRETRAINED_MODELS:
            raise ValueError("Invalid pretrained model {}, must be one of {}".format(
                args.pretrained_model, DEFAULT_PRETRAINED_MODELS))
        weights = load_weights(args.pretrained_model)
        args.read_proba_threshold = PRETRAINED_CONFIGS[args.pretrained_model][1]
        args.norm_path = PRETRAINED_CONFIGS[args.pretrained_model][2]

    if args.gpus < 1:
        raise ValueError("--gpus must be >= 1")
    if "M6A_RANK" in os.environ:                         # one rank of a --gpus N job (started by multi_gpu.launch)
        from .. import multi_gpu
        multi_gpu.run_rank(args, weights)
        return
    if args.gpus > 1:
        from .. import multi_gpu
        pathlib.Path(args.out_dir).mkdir(parents=True, exist_ok=True)
        rc = multi_gpu.launch(args, weights)
        if rc != 0:
            raise SystemExit(rc)
        return

    # the GPU context (HIP initialisation, weights) comes up on a thread while the loader parses data.json
    import threading
    made = {}

    def make_engine():
        try:
            made["engine"] = make_engine_for(args, weights, _device_index(args.device))
            made["engine"].prepare_host_io()     # pinned staging for the host arrays the loader is producing
        except BaseException as exc:        # re-raised on the main thread below
            made["error"] = exc

    starter = threading.Thread(target=make_engine)
    starter.start()
    pathlib.Path(args.out_dir).mkdir(parents=True, exist_ok=True)
    with open(os.path.join(args.out_dir, "data.site_proba.csv"), "w", encoding="utf-8") as f:
        f.write(SITE_HEADER)
    with open(os.path.join(args.out_dir, "data.indiv_proba.csv"), "w", encoding="utf-8") as g:
        g.write(INDIV_HEADER)
    # --n_processes is the reference's host-parallelism flag: here it sizes the loader / writer threads
    if any(str(d).endswith(STORE_SUFFIX) for d in args.input_dir):
        if len(args.input_dir) != 1:
            raise ValueError("--input_dir takes ONE site store (pack replicates together with `m6anet_amd pack`)")
        batch = open_store(args.input_dir[0], args.norm_path, DEFAULT_MIN_READS)
    else:
        try:
            batch = load_sites_native(args.input_dir, DEFAULT_MIN_READS, args.norm_path, n_threads=args.n_processes)
        except ImportError:
            batch = load_sites(args.input_dir, DEFAULT_MIN_READS, args.norm_path)   # libm6a_io.so not built
    starter.join()
    if "error" in made:
        raise made["error"]
    engine = made["engine"]
    run_inference(engine, batch, args)
    engine.close()
