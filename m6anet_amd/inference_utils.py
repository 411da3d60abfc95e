"""run_inference for the HIP engine: the host-side mirror of m6anet/utils/inference_utils.py.

The reference streams batches through a DataLoader and appends to the two CSVs per flush group
(inference_utils.py:33-67).  Here the whole job is one `engine.infer` call (the flush-group
geometry still decides every random draw -- it is passed down as batch_size/save_per_batch) and the
CSVs are written once, in the reference's exact row formats:

  data.site_proba.csv   '%s,%d,%s,%.16f,%s,%.16f'  tx_id, tx_pos, n_reads, site_prob, kmer, mod_ratio
  data.indiv_proba.csv  '%s,%d,%s,%.16f'           tx_id, tx_pos, read_id, read_prob

Deliberate divergence: the reference's inverted flush test (`if (it + 1) % save_per_batch`,
inference_utils.py:47) never writes the batches after the last flush (every run with an even number
of batches loses its final batch); every site is written here by default.  `--drop_unflushed_tail`
(args.drop_unflushed_tail) is the reference-compatible mode: the CSVs then hold exactly the reference's rows.
"""
import os

import numpy as np

from .constants import N_SAMPLES
from .engine import reference_written_sites

SITE_HEADER = "transcript_id,transcript_position,n_reads,probability_modified,kmer,mod_ratio\n"
INDIV_HEADER = "transcript_id,transcript_position,read_index,probability_modified\n"


def calculate_site_proba(engine, read_probs, n_iters, n_samples=N_SAMPLES, n_processes=1, seed=0):
    """Same call shape as the reference's calculate_site_proba(read_probs, n_iters, n_samples,
    n_processes) (inference_utils.py:90-104): `read_probs` is the list of per-site arrays of ONE
    flush group; returns the list of site probabilities.  n_processes is accepted and ignored -- the
    result is the reference's n_processes=1 stream."""
    del n_processes
    off = np.concatenate([[0], np.cumsum([len(p) for p in read_probs])]).astype(np.int64)
    p = np.concatenate(read_probs).astype(np.float32)
    site, _ = engine.calculate_site_proba(p, off, n_iters, n_samples, 0.0, seed, batch_size=max(len(read_probs), 1))
    return list(site)


def format_site_rows(batch, site_prob, mod_ratio, n_sites=None):
    n_reads = batch.n_reads
    return ["%s,%d,%s,%.16f,%s,%.16f\n" % (batch.tx_ids[s], batch.tx_pos[s], n_reads[s], site_prob[s],
                                           batch.kmer5[s], mod_ratio[s])
            for s in range(batch.n_sites if n_sites is None else n_sites)]


def format_indiv_rows(batch, read_prob, n_sites=None):
    rows = []
    off = batch.off
    for s in range(batch.n_sites if n_sites is None else n_sites):
        tx, pos, ids = batch.tx_ids[s], batch.tx_pos[s], batch.read_ids[s]
        p = read_prob[off[s]:off[s + 1]]
        rows.extend("%s,%d,%s,%.16f\n" % (tx, pos, ids[i], p[i]) for i in range(len(ids)))
    return rows


def run_inference(engine, batch, args):
    """engine: M6ANetEngine; batch: data_utils.SiteBatch; args: namespace with out_dir,
    num_iterations, batch_size, save_per_batch, seed, read_proba_threshold (scripts/inference.py)."""
    read_prob, site_prob, mod_ratio = engine.infer(
        batch.X, batch.site_kmers, batch.off, args.num_iterations, N_SAMPLES, args.read_proba_threshold,
        args.seed, args.batch_size, args.save_per_batch)
    n_write = None
    if getattr(args, "drop_unflushed_tail", False):       # the reference's row set (inference_utils.py:47)
        n_write = reference_written_sites(batch.n_sites, args.batch_size, args.save_per_batch)
    if batch.native is not None:          # native writer: same bytes, formatted on all host threads
        batch.native.write_csv(args.out_dir, read_prob, site_prob, mod_ratio, write_header=False, n_sites=n_write)
        return read_prob, site_prob, mod_ratio
    with open(os.path.join(args.out_dir, "data.site_proba.csv"), "a", encoding="utf-8") as f:
        f.writelines(format_site_rows(batch, site_prob, mod_ratio, n_write))
    with open(os.path.join(args.out_dir, "data.indiv_proba.csv"), "a", encoding="utf-8") as g:
        g.writelines(format_indiv_rows(batch, read_prob, n_write))
    return read_prob, site_prob, mod_ratio
