"""Stream sharding for multi-GPU runs.  Channel-streams are independent (R/enc_lc3_fl.c:167-171), so ranks simply own
disjoint contiguous blocks of streams for their whole lifetime: no data-path collective exists on this path."""


def stream_block(rank, world, n_streams_total):
    """[first, last) global stream indices owned by `rank` (contiguous, sizes differ by at most one)."""
    base, rem = divmod(n_streams_total, world)
    first = rank * base + min(rank, rem)
    return first, first + base + (1 if rank < rem else 0)


def owner_of(stream, world, n_streams_total):
    base, rem = divmod(n_streams_total, world)
    cut = rem * (base + 1)
    return stream // (base + 1) if stream < cut else rem + (stream - cut) // max(base, 1)
