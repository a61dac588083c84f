"""Build container only (no GPU): how much of the pinned lit frames' cone-trace work is spent on rays a min-distance brick table PROVES
open?  The CPU oracle marches every traced pair of the frame (the frames of tests/golden/full_frame_bands.json: cfg3 and
cfg5 as bench.py times them) and oracle/ilm_oracle_census.c evaluates the proof beside the march; every proven ray is checked to have
returned exactly 1.0f.  The decision rule of the review: build the early-out only if provably open rays carry >= 20 % of cfg5's samples.

    python tools/open_ray_census.py [cfg3|cfg5|both] [brick_texels ...]  > profiles/r05_open_ray_census.txt
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from illuminant_amd import scenes  # noqa: E402
from oracle import oracle as orc  # noqa: E402
import make_full_frame_bands as gen  # noqa: E402


def census(name, brick_texels, rows=None):
    w, h, dfu, lights, atlas, sfmt, garr, gfmt = gen.scene(name)
    env = scenes.environment(gbuffer_size=(w, h))
    tex, gtex = orc.make_texture(atlas, sfmt), orc.make_texture(garr, gfmt)
    c = orc.OpenRayCensus()
    t0 = time.time()
    b0, b1 = rows if rows else (0, h)
    for y in range(b0, b1, 128):
        orc.open_ray_census(lights, env, dfu, gtex, tex, w, h, y, min(b1, y + 128), brick_texels, c)
        print("  %s brick %d rows %d: %.0f s" % (name, brick_texels, y, time.time() - t0), file=sys.stderr, flush=True)
    return c.as_dict(), time.time() - t0


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    bricks = [int(v) for v in sys.argv[2:]] or [8]
    pinned = json.load(open(os.path.join(ROOT, "tests", "golden", "full_frame_bands.json")))
    print("open-ray census: rays of the pinned lit frames that a per-brick minimum-distance table PROVES unoccluded (coneTrace returns exactly 1.0f)")
    print("generator: tools/open_ray_census.py + oracle/ilm_oracle_census.c; frames: tests/golden/make_full_frame_bands.py scene()")
    for name in (("cfg3", "cfg5") if which == "both" else (which,)):
        for bt in bricks:
            c, secs = census(name, bt)
            tp, ts = c["traced_pairs"], c["traced_samples"]
            print("\n%s, bricks of %d x %d texels (+ 1 texel of apron), all slices folded; %.0f s of oracle time" % (name, bt, bt, secs))
            print("  traced pairs %d (fixture: %d)   cone-trace samples %d (the frame's %d include the AO samples)" % (tp, pinned[name]["traced"], ts, pinned[name]["sdf_samples"]))
            assert tp == pinned[name]["traced"], "the census marched another frame than the fixture pins"

            def pct(a, b):
                return "%5.1f %%" % (100.0 * a / max(b, 1))
            print("  marches that returned exactly 1.0f (ceiling of any exact early-out): %s of pairs, %s of samples" % (pct(c["result_one_pairs"], tp), pct(c["result_one_samples"], ts)))
            print("  PROVEN open, strict (visibility never leaves 1):                    %s of pairs, %s of samples" % (pct(c["strict_pairs"], tp), pct(c["strict_samples"], ts)))
            print("  PROVEN open, loose (every quotient >= 0.9501):                      %s of pairs, %s of samples" % (pct(c["loose_pairs"], tp), pct(c["loose_samples"], ts)))
            print("  proven rays whose march did NOT return exactly 1.0f: %d (must be 0)" % c["violations"])
            assert c["violations"] == 0
            print("  waves (8 x 8 pixels x one light) with a traced lane: %d; ALL traced lanes proven: %s of waves, carrying %s of samples"
                  % (c["wave_count"], pct(c["wave_open"], c["wave_count"]), pct(c["wave_open_samples"], ts)))
            print("  wave loop iterations (longest lane per wave) %d -> %d with the proven lanes removed: %s fewer"
                  % (c["wave_iterations"], c["wave_iterations_left"], pct(c["wave_iterations"] - c["wave_iterations_left"], c["wave_iterations"])))
            print("  bricks the proofs visited: %.1f per traced pair" % (c["dda_bricks"] / max(tp, 1)))
            print("  raw:", json.dumps(c))


if __name__ == "__main__":
    main()
