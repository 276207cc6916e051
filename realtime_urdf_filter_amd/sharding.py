"""Stream sharding across GPUs: streams are independent, so a batch is
partitioned with no data-path collective.  One process per GPU; the only collectives are the
barrier around the timed region and the tiny end-of-batch gather of per-rank results."""


def shard_range(n_items, world_size, rank):
    """Contiguous block partition of range(n_items): returns (first, count) of `rank`'s shard.
    The first n_items % world_size ranks get one extra item."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad world_size/rank %r/%r" % (world_size, rank))
    base, extra = divmod(n_items, world_size)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def models_for_rank(n_models, world_size, rank):
    """C5: URDF m lives on GPU m % world_size, so each GPU holds only its own models."""
    return [m for m in range(n_models) if m % world_size == rank]


def gather_frame_counts(dist, frames, elapsed_s, device=None):
    """All-gather {frames, elapsed} (a few bytes per rank) -> (total_frames, max_elapsed)."""
    import torch
    t = torch.tensor([float(frames), float(elapsed_s)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return int(sum(o[0].item() for o in out)), max(o[1].item() for o in out)
