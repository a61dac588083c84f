"""Multi-GPU partitioning of the two hot paths (one process per GPU, torch.distributed over RCCL / xGMI).

The reference is single-GPU: this is new host logic, kept free of device code so the
N > 1 path can be exercised on CPU with the `gloo` backend (tests/test_sharding_gloo.py).

Particles (ParticleSystem.cs:743-745: chunks never interact)
    chunk -> rank by `chunk_index mod world`: no data-path collective per step.  Only the per-chunk live counts
    (4 bytes each) are all-gathered, every liveness interval, so that every rank can run the reference's
    reap / LiveCount bookkeeping (ParticleLiveness.cs:80-129) on the whole table.
Lighting (every pixel needs the whole SDF atlas and every light: both are replicated)
    the frame is cut into `world` strips of whole 16-row tile bands; each rank renders its strip into its slice
    of a full-frame lightmap and the strips are all-gathered in place.  xGMI is point-to-point (7 links per GPU):
    an all-gather of 8 equal 8.3 MB strips is one transfer per link, so equal strips keep every link equally busy;
    `balanced_row_strips` trades that for equal render cost when the lights are unevenly spread.
"""
import numpy as np

TILE_ROWS = 16   # the sphere-light kernel works on 16x16 pixel tiles (csrc/lighting.hip)


# ---- particles -----------------------------------------------------------------------------------------------

def chunk_owner(chunk_index, world):
    return chunk_index % world


def owned_chunks(chunk_count, rank, world):
    """Global chunk indices rank `rank` owns, ascending."""
    return list(range(rank, chunk_count, world))


def gather_live_counts(local_counts, chunk_count, rank, world, dist, device=None):
    """All-gather of per-chunk live counts.  `local_counts[i]` belongs to global chunk owned_chunks(...)[i].
    Returns a numpy uint32 array over the global chunk table, identical on every rank (bit-exact: integers)."""
    import torch
    per_rank = (chunk_count + world - 1) // world
    mine = torch.zeros(per_rank, dtype=torch.int64, device=device)
    local = np.asarray(local_counts, dtype=np.int64)
    assert local.shape[0] == len(owned_chunks(chunk_count, rank, world))
    mine[:local.shape[0]] = torch.from_numpy(local).to(mine.device)
    if dist is None or world == 1:
        gathered = mine.reshape(1, per_rank)
    else:
        out = torch.empty(world * per_rank, dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(out, mine)
        gathered = out.reshape(world, per_rank)
    g = gathered.cpu().numpy()
    counts = np.zeros(chunk_count, dtype=np.uint32)
    for r in range(world):
        idx = owned_chunks(chunk_count, r, world)
        counts[idx] = g[r, :len(idx)]
    return counts


# ---- lighting ------------------------------------------------------------------------------------------------

def row_strips(height, world, align=TILE_ROWS):
    """`world` contiguous [begin, end) row ranges covering [0, height): equal numbers of `align`-row bands (the
    last strip takes the ragged remainder; strips may be empty when world > number of bands)."""
    bands = (height + align - 1) // align
    strips = []
    for r in range(world):
        b0 = (bands * r) // world
        b1 = (bands * (r + 1)) // world
        strips.append((min(b0 * align, height), min(b1 * align, height)))
    return strips


def padded_row_strips(height, world, align=TILE_ROWS):
    """Equal strips of R rows (R a multiple of `align`, world * R >= height) for the single in-place all-gather:
    returns (R, strips) where strips[r] = rows rank r renders, clipped to the frame.  The lightmap every rank
    holds is world * R rows tall (the frame is its first `height` rows), so every rank sends the same number of
    bytes over each xGMI link."""
    per = (height + world - 1) // world
    R = ((per + align - 1) // align) * align
    return R, [(min(r * R, height), min((r + 1) * R, height)) for r in range(world)]


def light_row_cost(lights, height):
    """Per-row render cost estimate: for every light, the width of its raster footprint
    (radius + ramp + 1, SphereLightCore.fxh:28-55) on each row it touches."""
    cost = np.zeros(height, dtype=np.float64)
    for L in lights:
        cy = float(L.LightPosition1.y)
        reach = float(L.LightProperties.x) + float(L.LightProperties.y) + 1.0
        y0 = max(0, int(np.floor(cy - reach)))
        y1 = min(height, int(np.ceil(cy + reach)))
        if y1 > y0:
            cost[y0:y1] += 2.0 * reach
    return cost


def balanced_row_strips(height, world, lights, align=TILE_ROWS):
    """Strips of whole `align`-row bands whose estimated costs are as equal as a greedy prefix split gives."""
    cost = light_row_cost(lights, height) + 1.0           # +1: the G-buffer read / lightmap write every row pays
    bands = (height + align - 1) // align
    band_cost = np.array([cost[b * align:min((b + 1) * align, height)].sum() for b in range(bands)])
    prefix = np.concatenate([[0.0], np.cumsum(band_cost)])
    total = prefix[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        b = int(np.searchsorted(prefix, target, side="left"))
        if b > 0 and abs(prefix[b - 1] - target) <= abs(prefix[min(b, bands)] - target):
            b -= 1
        b = min(max(b, cuts[-1]), bands)
        cuts.append(b)
    cuts.append(bands)
    return [(min(cuts[r] * align, height), min(cuts[r + 1] * align, height)) for r in range(world)]


def rebalance_row_strips(strips, seconds, height, align=TILE_ROWS):
    """The strips cut again from what they COST: `seconds[r]` = the time rank r's strip took (any unit).  The footprint model of
    balanced_row_strips cannot see the obstacle field (a ray in the open takes a few long steps, one along a wall dozens) and leaves the
    strips of cfg5 7-9 % apart; a frame or two of measured times bring them within ~2 %.  The cost of a row is taken as constant inside
    its current strip (time / rows); the new cuts are where the cumulative cost passes r / world of the total, on `align`-row bands.
    Every rank must pass the same `seconds` (gather them first): the table is part of the exchange protocol."""
    world = len(strips)
    assert len(seconds) == world
    bands = (height + align - 1) // align
    band_cost = np.zeros(bands, dtype=np.float64)
    for (b, e), t in zip(strips, seconds):
        rows = max(e - b, 0)
        if rows == 0:
            continue
        per_row = max(float(t), 0.0) / rows
        for band in range(b // align, (e + align - 1) // align):
            r0, r1 = max(b, band * align), min(e, (band + 1) * align)
            band_cost[band] += per_row * max(r1 - r0, 0)
    if not (band_cost.sum() > 0):
        return list(strips)
    prefix = np.concatenate([[0.0], np.cumsum(band_cost)])
    total = prefix[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        b = int(np.searchsorted(prefix, target, side="left"))
        if b > 0 and abs(prefix[b - 1] - target) <= abs(prefix[min(b, bands)] - target):
            b -= 1
        b = min(max(b, cuts[-1]), bands)
        cuts.append(b)
    cuts.append(bands)
    return [(min(cuts[r] * align, height), min(cuts[r + 1] * align, height)) for r in range(world)]


def all_gather_rows(full, strips, rank, dist):
    """In-place all-gather of row strips of `full` (a (H, W, C) torch tensor every rank holds; rank r has rendered
    rows strips[r]).  Equal strips use one all_gather_into_tensor straight into `full`; unequal strips are
    broadcast strip by strip (same bytes on the wire, `world` smaller collectives)."""
    world = len(strips)
    if dist is None or world == 1:
        return full
    if full.shape[0] % world == 0:
        R = full.shape[0] // world
        if all((e == b) or (b == r * R and e <= (r + 1) * R) for r, (b, e) in enumerate(strips)):
            # equal slots (padded_row_strips, or a height that divides evenly): ONE in-place all-gather; rank r's
            # input is the r-th slot of the output, which is the in-place form RCCL supports
            mine = full[rank * R:(rank + 1) * R]
            dist.all_gather_into_tensor(full, mine.clone() if full.device.type == "cpu" else mine)
            return full
    for r, (b, e) in enumerate(strips):
        if e > b:
            dist.broadcast(full[b:e], src=r)
    return full


def exchange_row_ranges(full, strips, rank, dist):
    """The exchange csrc/group.hip exchange_ranges performs for UNEQUAL strips (ilm_group_lightmap_set_strips): rank r sends its rows
    strips[r] to each of the world - 1 other ranks and receives theirs at their true rows -- point-to-point, staggered so that no two
    ranks start on the same peer (peer k steps ahead / behind), one transfer per link and direction on a fully connected mesh."""
    world = len(strips)
    if dist is None or world == 1:
        return full
    b, e = strips[rank]
    ops = []
    for k in range(1, world):
        to, frm = (rank + k) % world, (rank - k + world) % world
        if e > b:
            ops.append(dist.P2POp(dist.isend, full[b:e].contiguous(), to))
        fb, fe = strips[frm]
        if fe > fb:
            ops.append(dist.P2POp(dist.irecv, full[fb:fe], frm))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    return full
