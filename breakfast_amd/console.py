"""CLI shell equivalent to the reference's `breakfast` command (src/breakfast/console.py:10-170):
same options, defaults, non-DNA overrides and parameter echo; the clustering step runs on the GPU."""

import os
import pathlib

import click
from click.core import ParameterSource

from . import __version__

_DNA = ("covsonar_dna", "nextclade_dna")


@click.command(context_settings={"show_default": True})
@click.option("--input-file", type=click.Path(exists=True, path_type=pathlib.Path), required=True, help="Input file")
@click.option("--sep", default="\t", help="Input file separator")
@click.option("--outdir", type=click.Path(path_type=pathlib.Path), default="output",
              help="Output directory for all output files")
@click.option("--max-dist", type=click.IntRange(0), default=1, help="Maximum parwise distance")
@click.option("--min-cluster-size", type=click.IntRange(1), default=2, help="Minimum cluster size")
@click.option("--input-cache", type=click.Path(exists=True, path_type=pathlib.Path),
              help="Input cached pickle file from previous run")
@click.option("--output-cache", type=click.Path(path_type=pathlib.Path), help="Path to Output cached pickle file")
@click.option("--id-col", default="accession", help="Column with the sequence identifier")
@click.option("--clust-col", default="dna_profile", help="Metadata column to cluster")
@click.option("--var-type", default="covsonar_dna", help="Type of variants",
              type=click.Choice(["covsonar_dna", "covsonar_aa", "nextclade_dna", "nextclade_aa", "raw"]))
@click.option("--sep2", default=" ", help="Secondary clustering column separator (between each mutation)")
@click.option("--trim-start", type=click.IntRange(0), default=264, help="Bases to trim from the beginning (0 = disable)")
@click.option("--trim-end", type=click.IntRange(0), default=228, help="Bases to trim from the end (0 = disable)")
@click.option("--reference-length", type=click.IntRange(0), default=29903,
              help="Length of reference genome (defaults to NC_045512.2 length)")
@click.option("--skip-del/--no-skip-del", default=True, help="Skip deletions")
@click.option("--skip-ins/--no-skip-ins", default=True, help="Skip insertions")
@click.option("--jobs", type=click.IntRange(1), default=1, envvar="OMP_NUM_THREADS",
              help="Number of jobs (threads); kept for compatibility, the GPU path ignores it")
@click.option("--gpus", type=click.IntRange(1), default=1, envvar="BFK_GPUS",
              help="GPUs to shard the distance work over (one process, one context per device; not in the reference)")
@click.version_option(version=__version__)
def main(input_file, outdir, input_cache, output_cache, id_col, clust_col, var_type, sep, sep2, max_dist,
         min_cluster_size, trim_start, trim_end, reference_length, skip_del, skip_ins, jobs, gpus):
    if var_type not in _DNA:
        # trimming / indel skipping only make sense for DNA profiles: explicit requests are errors,
        # the DNA-oriented defaults are switched off
        ctx = click.get_current_context()
        src = ctx.get_parameter_source
        if trim_start != 0 and src("trim_start") != ParameterSource.DEFAULT:
            raise click.BadParameter("Can not trim non-DNA features")
        if trim_end != 0 and src("trim_end") != ParameterSource.DEFAULT:
            raise click.BadParameter("Can not trim non-DNA features")
        if skip_del and src("skip_del") == ParameterSource.COMMANDLINE:
            raise click.BadParameter("Can not skip indels in non-DNA features")
        if skip_ins and src("skip_ins") == ParameterSource.COMMANDLINE:
            raise click.BadParameter("Can not skip indels in non-DNA features")
        trim_start = trim_end = 0
        skip_del = skip_ins = False
    if trim_start > reference_length or trim_end > reference_length:
        raise click.BadParameter("Can not trim more than the reference length")

    print("Clustering sequences")
    for label, value in (
        ("Input file", input_file), ("Input file separator", f"'{sep}'"), ("ID column", id_col),
        ("clustering feature type", var_type), ("clustering feature column", clust_col),
        ("clustering feature column separator", f"'{sep2}'"), ("max dist", max_dist),
        ("minimum cluster size", min_cluster_size), ("trim start (bp)", trim_start), ("trim end (bp)", trim_end),
        ("reference length (bp)", reference_length), ("skip deletions", skip_del), ("skip insertions", skip_ins),
        ("Input cache file", input_cache), ("Output cache file", output_cache),
    ):
        print(f"  {label} = {value}")
    os.environ["OMP_NUM_THREADS"] = str(jobs)

    # native end-to-end path (reader, filter + collapse + CSR, GPU clustering, writer: fastpath.py); it declines
    # inputs that need pandas' CSV dialect handling, the reference's exceptions, or the cache
    if os.environ.get("BFK_NO_FASTPATH") != "1":
        from . import fastpath

        if fastpath.run(input_file, sep, id_col, clust_col, var_type, sep2, skip_ins, skip_del, trim_start, trim_end,
                        reference_length, max_dist, min_cluster_size, outdir, input_cache, output_cache, gpus):
            return
    from . import breakfast  # pandas-based mirror of the reference's functions

    meta = breakfast.read_input(input_file, sep, id_col, clust_col)
    meta["feature"] = breakfast.filter_features(meta["feature"], sep2, var_type, skip_ins, skip_del, trim_start,
                                                trim_end, reference_length)
    meta_nodups = breakfast.collapse_duplicates(meta)
    meta_clustered = breakfast.cluster(meta_nodups, sep2, max_dist, min_cluster_size, input_cache, output_cache, n_gpus=gpus)
    breakfast.write_output(meta_clustered, meta, outdir)
