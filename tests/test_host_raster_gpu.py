"""Host mirror: ParticleSystem.Render (ParticleSystem.cs:943-1041) driving ilm_render_particles, against the oracle's rasteriser on the
state read back from the system."""
import numpy as np
import pytest

from illuminant_amd import abi, scenes
from tests.test_raster_gpu import compare_images

pytestmark = pytest.mark.gpu

P, V, A, RC, RD = abi.PLANE_POSITION, abi.PLANE_VELOCITY, abi.PLANE_ATTRIBUTES, abi.PLANE_RENDER_COLOR, abi.PLANE_RENDER_DATA


def test_render_after_updates_matches_oracle(oracle):
    from illuminant_amd import _host as H
    ctx = H.DeviceContext(0)
    cs = 64
    tp = H.ManualTimeProvider()
    ecfg = H.ParticleEngineConfiguration(cs)
    ecfg.TimeProvider = tp
    engine = H.ParticleEngine(ctx, ecfg, scenes.randomness_table(7))
    cfg = H.ParticleSystemConfiguration()
    cfg.LifeDecayPerSecond = 0.5
    cfg.Size = [5.0, 3.0]
    cfg.RotationFromLife = 90.0                      # degrees per unit of life: sprites at every angle
    cfg.ZToY = 0.5
    cfg.SizeFromZ = 0.02
    col = H.ParticleColor(); col.Global = [0.9, 0.7, 1.0, 0.6]
    cfg.Color = col
    ap = H.ParticleAppearance(); ap.Rounded = True
    cfg.Appearance = ap
    ps = H.ParticleSystem(engine, cfg)
    sp = H.Spawner(11)
    sp.MinRate = sp.MaxRate = 9000.0
    f = H.Formula3(); f.Constant = [160, 100, 4]; f.RandomScale = [150, 90, 4]; f.Type = H.FormulaType.Spherical
    sp.Position = f
    g = H.Formula3(); g.RandomScale = [40, 40, 0]; g.Type = H.FormulaType.Spherical
    sp.Velocity = g
    life = H.Formula1(); life.Constant = 2.0; life.RandomScale = 2.0
    sp.Life = life
    c4 = H.Formula4(); c4.Constant = [0.6, 0.5, 0.4, 0.5]; c4.RandomScale = [0.4, 0.5, 0.6, 0.5]
    sp.Color = c4
    ps.AddTransform(sp)
    for frame in range(30):
        tp.Advance(1.0 / 60.0)
        ps.Update(frame)
    w, h = 320, 200
    target = H.RenderTarget(ctx, w, h)
    clear = [0.02, 0.03, 0.04, 1.0]
    target.Clear(clear)
    origin, scale, vscale, vpos = [4.0, -3.0], [1.0, 1.0], [1.0, 1.0], [2.0, 1.0]
    live, pairs, shaded = ps.Render(target, abi.BLEND_ALPHA, origin, scale, vscale, vpos, True)
    got = target.Download()
    # the oracle on the same state, with the parameter block the mirror built
    params = abi.RasterizeParams.from_buffer_copy(ps.RasterizeParamsBytes(abi.BLEND_ALPHA, origin, scale, vscale, vpos))
    assert params.RenderingOptions[0] == 1.0 and abs(params.GlobalColor.x - 0.9 * 0.6) < 1e-6 and params.SystemSize[0] == 5.0
    chunks = [[ps.Readback(ci, k) for k in (P, V, A, RC, RD)] for ci in range(len(ps.Chunks))]
    quads = [min(cs * cs, c.TotalSpawned + 1) for c in ps.Chunks]
    want = np.zeros((h, w, 4), np.float32); want[:] = clear
    want, (olive, oshaded) = oracle.render_particles(chunks, params, w, h, quad_counts=quads, image=want)
    assert live == olive and live > 3000 and abs(shaded - oshaded) <= 8 and pairs >= live
    compare_images(got, want, "host render", max_outliers=8)
    # a second Render blends on top of the first (no implicit clear)
    ps.Render(target, abi.BLEND_ADDITIVE, origin, scale, vscale, vpos, False)
    again = target.Download()
    assert (again >= got - 1e-6).all() and (again > got + 1e-3).mean() > 0.3
    # a textured system: Appearance.Texture (8 x 8 frames of a 32 x 16 sheet), point sampled, frames from life
    sheet = scenes.uniform(77, (16, 32, 4), 0.0, 1.0)
    ap.TextureSize = [32.0, 16.0]; ap.SizePx = [8.0, 8.0]; ap.Bilinear = False; ap.AnimationRate = [0.5, 0.0]
    cfg.Appearance = ap
    ps.Configuration = cfg
    with pytest.raises(H.NativeException):
        ps.Render(target)                              # ILM_ERR_STATE: no bitmap bound yet
    ps.SetBitmap(sheet)
    target.Clear(clear)
    live2, _, shaded2 = ps.Render(target, abi.BLEND_ALPHA, origin, scale, vscale, vpos, True)
    got2 = target.Download()
    params2 = abi.RasterizeParams.from_buffer_copy(ps.RasterizeParamsBytes(abi.BLEND_ALPHA, origin, scale, vscale, vpos))
    assert params2.BitmapFilter == abi.BITMAP_POINT and params2.AnimationRate[0] == 2.0 and params2.SizeFactorAndPosition.x == 4.0
    want2 = np.zeros((h, w, 4), np.float32); want2[:] = clear
    want2, (olive2, oshaded2) = oracle.render_particles(chunks, params2, w, h, quad_counts=quads, image=want2, bitmap=sheet)
    assert live2 == olive2 and abs(shaded2 - oshaded2) <= 8
    compare_images(got2, want2, "host render, textured", max_outliers=40)
