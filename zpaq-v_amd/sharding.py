"""Static round-robin sharding of independent ZPAQ blocks over GPUs.

Blocks share nothing but read-only tables (every block starts from a fresh
Predictor/ZPAQL: reference zpaq/compressor.v:90,147-148,184-185), so block b goes
to rank b mod world and no data-path collective exists.  The only
cross-rank step is the host putting results back in block order.
"""


def shard_indices(nblocks_total, rank, world):
    """Global block ids owned by `rank`: b = rank, rank+world, ..."""
    return list(range(rank, nblocks_total, world))


def merge_in_block_order(per_rank_results, nblocks_total):
    """per_rank_results[r] = list of results for shard_indices(nblocks_total, r, world)."""
    world = len(per_rank_results)
    out = [None] * nblocks_total
    for r, res in enumerate(per_rank_results):
        ids = shard_indices(nblocks_total, r, world)
        assert len(ids) == len(res), (r, len(ids), len(res))
        for b, x in zip(ids, res):
            out[b] = x
    return out
