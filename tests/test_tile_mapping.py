"""The light pass's block -> tile mapping (lighting.hip sphere_lights_kernel, tile_map 4; launch_sphere_lights_prepared sizes the grid):
square groups of M x M tiles dealt round-robin to the eight XCDs (block b runs on XCD b % 8).  The arithmetic restated here must send
exactly one block to every tile of any frame -- the GPU test test_every_pixel_of_any_frame_size_is_rendered_once checks the kernel
against the same property on the device."""
import numpy as np
import pytest


def grid_blocks(tiles_x, tiles_y, M):
    groups = ((tiles_x + M - 1) // M) * ((tiles_y + M - 1) // M)
    return ((groups + 7) // 8) * 8 * M * M


def tile_of_block(b, tiles_x, tiles_y, M):
    xcd, k = b % 8, b // 8
    g, t = (k // (M * M)) * 8 + xcd, k % (M * M)
    mx = (tiles_x + M - 1) // M
    ty, tx = (g // mx) * M + t // M, (g % mx) * M + t % M
    return ty * tiles_x + tx if (tx < tiles_x and ty < tiles_y) else None


@pytest.mark.parametrize("M", [1, 2, 3, 6, 8, 16])
def test_every_tile_gets_exactly_one_block(M):
    rng = np.random.default_rng(M)
    sizes = [(1, 1), (6, 6), (7, 5), (12, 13), (120, 68), (240, 135), (240, 17), (5, 240)] + [tuple(int(v) for v in rng.integers(1, 90, 2)) for _ in range(20)]
    for tiles_x, tiles_y in sizes:
        seen = np.zeros(tiles_x * tiles_y, np.int32)
        for b in range(grid_blocks(tiles_x, tiles_y, M)):
            t = tile_of_block(b, tiles_x, tiles_y, M)
            if t is not None:
                seen[t] += 1
        assert (seen == 1).all(), (tiles_x, tiles_y, M)


def test_an_xcd_walks_whole_groups_of_neighbours():
    """What the mapping is for: the 36 consecutive blocks an XCD receives cover one 6 x 6 square of tiles."""
    tiles_x, tiles_y, M = 240, 135, 6
    for xcd in range(8):
        for run in (0, 5, 57):
            tiles = [tile_of_block((run * 36 + i) * 8 + xcd, tiles_x, tiles_y, M) for i in range(36)]
            tiles = [t for t in tiles if t is not None]
            xs, ys = [t % tiles_x for t in tiles], [t // tiles_x for t in tiles]
            assert max(xs) - min(xs) < 6 and max(ys) - min(ys) < 6 and len(set(tiles)) == len(tiles)
