"""ilm_gbuffer_render_meshes on the MI355X against the oracle (oracle/ilm_oracle_gbuffer.c): 2.5D height volumes under the depth test,
billboards of both types with textures of every format, both G-buffer formats, and the lit frame through the generated G-buffer.
Coverage, depth decisions, relativeY, the encoded z and the normal's second component are pure IEEE arithmetic and must be bit-equal;
only the first component passes through atan2 (device libm vs glibc: a few ulp)."""
import numpy as np
import pytest

from illuminant_amd import abi, native, scenes

pytestmark = pytest.mark.gpu


def random_scene(seed, w, h, n_volumes=14, n_billboards=10, z_to_y=0.6):
    """volumes with random convex-ish / concave polygons, billboards of both types in SortKey order"""
    r = scenes.uniform(seed, (n_volumes, 16))
    volumes = []
    for v in range(n_volumes):
        cx, cy = 10 + r[v, 0] * (w - 20), 20 + r[v, 1] * (h - 30)
        n = 3 + int(r[v, 2] * 5)
        ang = np.sort(scenes.uniform(seed * 31 + v, (n,)) * 2 * np.pi)
        rad = 6 + scenes.uniform(seed * 57 + v, (n,)) * (10 + 30 * r[v, 3])
        poly = [(float(np.float32(cx + rad[k] * np.cos(ang[k]))), float(np.float32(cy + rad[k] * np.sin(ang[k])))) for k in range(n)]
        volumes.append((poly, float(np.float32(r[v, 4] * 10)), float(np.float32(4 + r[v, 5] * 50)), r[v, 6] < 0.8, r[v, 7] < 0.7))
    order = sorted(range(n_volumes), key=lambda i: -(np.float32(volumes[i][1]) + np.float32(volumes[i][2])))     # OrderByDescending, :248
    top = np.concatenate([scenes.top_face_mesh(volumes[i][0], volumes[i][1], volumes[i][2], volumes[i][3]) for i in order])
    front = np.concatenate([scenes.front_face_mesh(volumes[i][0], volumes[i][1], volumes[i][2], volumes[i][4]) for i in order] +
                           [np.zeros((0, 9), np.float32)])
    b = scenes.uniform(seed + 1000, (n_billboards, 16))
    boards, kinds = [], []
    for k in range(n_billboards):
        x, y = b[k, 0] * (w - 30), b[k, 1] * (h - 40)
        kind = abi.BILLBOARD_GBUFFER_DATA if b[k, 2] < 0.4 else abi.BILLBOARD_MASK
        d = dict(screen_bounds=((x, y), (x + 8 + b[k, 3] * 40, y + 8 + b[k, 4] * 50)), type=kind,
                 normal=(b[k, 5] - 0.5, b[k, 6], b[k, 7] * 0.5), cylinder_factor=(b[k, 8] if b[k, 9] < 0.5 else 0.0),
                 data_scale=(None if b[k, 10] < 0.3 else 0.25 + 4 * b[k, 10]), static_lighting_only=b[k, 11] < 0.3,
                 world_offset=(0.0, 0.0, b[k, 12] * 4), texture_bounds=((0.0, 0.0), (1.0, 1.0)) if b[k, 13] < 0.4 else (((0.25, 0.1), (0.9, 1.2)) if b[k, 13] < 0.7 else ((-0.4, -0.3), (1.5, 1.25))))
        if kind == abi.BILLBOARD_GBUFFER_DATA:
            d["world_elevation"] = b[k, 14] * 20
        if b[k, 15] < 0.3:
            d.pop("screen_bounds")
            d["world_bounds"] = ((x, y + 30, b[k, 14] * 30), (x + 20 + b[k, 3] * 30, y + 30, 0.0))
        boards.append(d)
        kinds.append(kind)
    return volumes, top, front, scenes.billboard_vertices(boards, 0.0, z_to_y), kinds


def textures_for(seed, kinds):
    """one texture per billboard (= per run): Color, HalfVector4, Vector4 and none, cycling"""
    out = []
    for k, kind in enumerate(kinds):
        tw, th = 5 + (k * 3) % 11, 4 + (k * 5) % 9
        t = scenes.uniform(seed + 17 * k, (th, tw, 4))
        t[..., 3] = np.where(scenes.uniform(seed + 17 * k + 1, (th, tw)) < 0.35, 0.0, t[..., 3] * 0.6 + 0.4)
        which = k % 4
        out.append(None if which == 3 else (np.round(t * 255).astype(np.uint8) if which == 0 else t.astype(np.float16) if which == 1 else t.astype(np.float32)))
    return out


def compare(got, want, fmt):
    """NaN normals (a g-data texel whose (r, g) lies outside the unit disc: sqrt of a negative, GBufferBitmap.fx:82) must be NaN in both"""
    if fmt == abi.GBUFFER_HALF4:
        g = got.view(np.float16).astype(np.float32)
        wv = want.astype(np.float16).astype(np.float32)
        tol = 2.0 ** -10                    # one half ulp below 1 when atan2's last bits straddle a rounding tie
    else:
        g, wv, tol = got, want, 1e-6
    assert np.array_equal(g[..., 1:], wv[..., 1:], equal_nan=True)
    assert np.array_equal(np.isnan(g[..., 0]), np.isnan(wv[..., 0]))
    assert np.nanmax(np.abs(g[..., 0] - wv[..., 0])) <= tol


@pytest.mark.parametrize("fmt", [abi.GBUFFER_FLOAT4, abi.GBUFFER_HALF4])
@pytest.mark.parametrize("seed", [3, 4])
def test_2p5d_scene_matches_oracle(ctx, oracle, fmt, seed):
    w, h = 200, 136
    z_to_y = 0.6
    _, top, front, bb, kinds = random_scene(seed, w, h, z_to_y=z_to_y)
    so, zso = scenes.self_occlusion_hacks(0.5, 64.0, 12)
    d = scenes.gbuffer_mesh_desc(ground_z=0.0, viewport_position=(3.0, -2.0), viewport_scale=(1.125, 1.25), z_to_y=z_to_y, render_scale=(1.0, 1.0),
                                 extent_z=64.0, self_occlusion_hack=so, z_self_occlusion_hack=zso, two_point_five_d=True)
    texs = textures_for(seed, kinds)
    handles = []
    for t in texs:
        if t is None:
            handles.append(None)
            continue
        fm = {np.dtype(np.uint8): abi.LIGHTMAP_RGBA8, np.dtype(np.float16): abi.LIGHTMAP_HALF4, np.dtype(np.float32): abi.LIGHTMAP_FLOAT4}[t.dtype]
        lm = native.Lightmap(ctx, t.shape[1], t.shape[0], fm)
        lm.upload(t)
        handles.append(lm)
    runs_o = [(q, 1, kinds[q]) for q in range(len(kinds))]
    runs_g = [(handles[q], q, 1, kinds[q]) for q in range(len(kinds))]
    gb = native.GBufferTexture(ctx, None, fmt, size=(w, h))
    gb.render_meshes(d, top, front, bb, runs_g)
    got = gb.download()
    want = oracle.render_gbuffer_meshes(w, h, d, top, front, bb, runs_o, texs)
    compare(got, want, fmt)
    # the scene exercises every shader: top and front faces, both billboard types, clipped tops (volumes above the 64-deep field)
    wz = want[..., 3]
    assert (np.abs(want[..., 1] - 0.5) < 1e-6).sum() > 200 and (want[..., 2] > 1.0).sum() > 2000 and (wz < 0).sum() > 100
    for x in handles:
        if x is not None:
            x.close()
    gb.close()


def test_1080p_frame_of_2500_triangles_matches_oracle(ctx, oracle):
    """The bench row's size (1920 x 1080, 256 volumes + 64 billboards): more triangles than one binning round holds per tile list
    refill, tiles crossed by hundreds of records, volumes taller than the field (clipped tops)."""
    w, h = 1920, 1080
    _, top, front, bb, kinds = random_scene(21, w, h, n_volumes=256, n_billboards=64, z_to_y=0.6)
    so, zso = scenes.self_occlusion_hacks(0.25, 128.0, 33)
    d = scenes.gbuffer_mesh_desc(z_to_y=0.6, extent_z=48.0, self_occlusion_hack=so, z_self_occlusion_hack=zso)
    texs = textures_for(21, kinds)
    handles = []
    for t in texs:
        if t is None:
            handles.append(None)
            continue
        fm = {np.dtype(np.uint8): abi.LIGHTMAP_RGBA8, np.dtype(np.float16): abi.LIGHTMAP_HALF4, np.dtype(np.float32): abi.LIGHTMAP_FLOAT4}[t.dtype]
        lm = native.Lightmap(ctx, t.shape[1], t.shape[0], fm)
        lm.upload(t)
        handles.append(lm)
    gb = native.GBufferTexture(ctx, None, abi.GBUFFER_FLOAT4, size=(w, h))
    gb.render_meshes(d, top, front, bb, [(handles[q], q, 1, kinds[q]) for q in range(len(kinds))])
    got = gb.download()
    want = oracle.render_gbuffer_meshes(w, h, d, top, front, bb, [(q, 1, kinds[q]) for q in range(len(kinds))], texs)
    compare(got, want, abi.GBUFFER_FLOAT4)
    assert (len(top) + len(front)) // 3 + 2 * len(kinds) > 2000
    assert (want[..., 2] > 1.0).mean() > 0.08                   # a tenth of the frame is volume surface
    for x in handles:
        if x is not None:
            x.close()
    gb.close()


def test_squares_crossed_by_thousands_of_triangles(ctx, oracle):
    """6000 small top-face triangles heaped on a 40 x 40 pixel patch (and a few far away) of a 144 x 80 frame: the patch's 64 x 64
    block list takes six binning rounds of 1024, every 8 x 8 square of the patch walks thousands of candidates 64 at a time, and the
    depth test (heights drawn in no order) must still see every triangle in draw order."""
    w, h = 144, 80
    n = 6000
    r = scenes.uniform(91, (n, 8))
    rows = []
    for t in range(n):
        far = (t % 97) == 0
        cx, cy = (100 + r[t, 0] * 40, 50 + r[t, 1] * 28) if far else (4 + r[t, 0] * 40, 4 + r[t, 1] * 40)
        z = float(np.float32(2 + r[t, 2] * 60))
        size = 2 + r[t, 3] * 14
        ang = r[t, 4] * 2 * np.pi + np.array([0.0, 2.1 + r[t, 5], 4.2 + r[t, 6]])
        for k in range(3):
            rows.append([cx + size * np.cos(ang[k]), cy + size * np.sin(ang[k]), z, 0.0, 0.0, 1.0, 0.0, z, 1.0 if r[t, 7] < 0.8 else 0.0])
    top = np.asarray(rows, np.float32)
    so, zso = scenes.self_occlusion_hacks(0.5, 64.0, 12)
    d = scenes.gbuffer_mesh_desc(z_to_y=0.25, extent_z=64.0, self_occlusion_hack=so, z_self_occlusion_hack=zso)
    gb = native.GBufferTexture(ctx, None, abi.GBUFFER_FLOAT4, size=(w, h))
    gb.render_meshes(d, top)
    got = gb.download()
    want = oracle.render_gbuffer_meshes(w, h, d, top)
    compare(got, want, abi.GBUFFER_FLOAT4)
    assert (want[:48, :48, 2] > 0.0).mean() > 0.5
    gb.close()


@pytest.mark.parametrize("budget", [1 << 20, 60000, 1])
def test_block_lists_over_larger_blocks_when_the_scratch_budget_is_small(ctx, oracle, budget, monkeypatch):
    """The binning pass's blocks double from 64 x 64 pixels until blocks x triangles x 4 B fit the budget (256 MB by default): with the
    budget turned down the same frame goes through 64-pixel blocks, 128-pixel blocks and one block for the whole frame -- same texels."""
    w, h = 520, 300
    z_to_y = 0.6
    _, top, front, bb, kinds = random_scene(31, w, h, n_volumes=40, n_billboards=12, z_to_y=z_to_y)
    so, zso = scenes.self_occlusion_hacks(0.5, 64.0, 12)
    d = scenes.gbuffer_mesh_desc(z_to_y=z_to_y, extent_z=64.0, self_occlusion_hack=so, z_self_occlusion_hack=zso)
    runs = [(None, q, 1, kinds[q]) for q in range(len(kinds))]
    want = oracle.render_gbuffer_meshes(w, h, d, top, front, bb, [(q, 1, kinds[q]) for q in range(len(kinds))])
    monkeypatch.setenv("ILM_GBUFFER_BLOCK_LIST_BYTES", str(budget))
    gb = native.GBufferTexture(ctx, None, abi.GBUFFER_FLOAT4, size=(w, h))
    gb.render_meshes(d, top, front, bb, runs)
    compare(gb.download(), want, abi.GBUFFER_FLOAT4)
    gb.close()


def test_non_2p5d_meshes_equal_the_polygon_entry_point(ctx, oracle):
    """ilm_gbuffer_render decides top-face coverage per pixel centre against the polygon; the mesh entry point rasterises a
    triangulation of it: same picture wherever no centre sits exactly on an edge (quarter-pixel vertices, unit scale)."""
    w, h = 160, 112
    volumes = [
        ([(10.25, 10.25), (70.25, 14.75), (64.5, 60.25), (30.25, 40.75), (12.75, 70.25)], 0.0, 24.0, True, True),
        ([(50.25, 30.75), (150.25, 30.75), (150.25, 100.25), (50.25, 100.25)], 6.0, 30.0, True, False),
        ([(100.75, 5.25), (140.25, 12.25), (120.25, 40.75)], 0.0, 12.0, True, True),
    ]
    vols, poly = scenes.height_volume_arrays(volumes)
    a = native.GBufferTexture(ctx, None, abi.GBUFFER_FLOAT4, size=(w, h))
    b = native.GBufferTexture(ctx, None, abi.GBUFFER_FLOAT4, size=(w, h))
    a.render(scenes.gbuffer_render_desc(0.0), vols, poly)
    order = sorted(range(len(volumes)), key=lambda i: volumes[i][1] + volumes[i][2])
    top = np.concatenate([scenes.top_face_mesh(volumes[i][0], volumes[i][1], volumes[i][2], volumes[i][4]) for i in order])
    d = scenes.gbuffer_mesh_desc(0.0, two_point_five_d=False)
    b.render_meshes(d, top)
    ga, gb_ = a.download(), b.download()
    assert np.array_equal(gb_, oracle.render_gbuffer_meshes(w, h, d, top))
    diff = np.any(ga != gb_, axis=-1)
    assert diff.sum() < 40                     # centres exactly on an edge (the KAT test enumerates them)
    a.close()
    b.close()


def test_lit_frame_through_the_2p5d_gbuffer(ctx, oracle):
    """The generated 2.5D G-buffer drives sampleGBuffer like an uploaded one (LightCommon.fxh:58-144: relativeY and z rebuild the
    shaded position, the spherical normal the N.L term)."""
    from tests.util import assert_close
    w, h = 160, 112
    z_to_y = 0.5
    _, top, front, bb, kinds = random_scene(9, w, h, n_volumes=8, n_billboards=4, z_to_y=z_to_y)
    so, zso = scenes.self_occlusion_hacks(0.5, 96.0, 12)
    d = scenes.gbuffer_mesh_desc(z_to_y=z_to_y, extent_z=96.0, self_occlusion_hack=so, z_self_occlusion_hack=zso)
    gb = native.GBufferTexture(ctx, None, abi.GBUFFER_FLOAT4, size=(w, h))
    runs = [(None, q, 1, kinds[q]) for q in range(len(kinds))]
    gb.render_meshes(d, top, front, bb, runs)
    got = gb.download()
    want = oracle.render_gbuffer_meshes(w, h, d, top, front, bb, [(q, 1, kinds[q]) for q in range(len(kinds))])
    compare(got, want, abi.GBUFFER_FLOAT4)
    layout = scenes.DistanceFieldLayout(256, 192, 96.0, 12, 0.5, 128)
    atlas = scenes.build_sdf_atlas(layout, scenes.random_obstacles(5, 14, (256, 192), size_lo=8.0, size_hi=30.0, z_hi=40.0))
    dfu = layout.uniforms(max_cone_radius=24.0, power=0.7, step_limit=64, min_step_size=1.0, long_step_factor=0.5)
    lights = scenes.random_lights(6, 6, w, h, z=(30.0, 60.0), radius=10.0, ramp=(40.0, 120.0))
    env = scenes.environment(z_to_y=z_to_y, gbuffer_size=(w, h))
    sdf = native.DistanceFieldTexture(ctx, atlas)
    lm = native.Lightmap(ctx, w, h, abi.LIGHTMAP_FLOAT4)
    native.render_sphere_lights(ctx, lights, env, dfu, gb, sdf, (0.05, 0.05, 0.05, 1.0), lm)
    lit = lm.download()
    # the oracle lights the DEVICE's G-buffer: the two differ in the last bits of the normal's first component only, and the light pass
    # is compared on equal inputs
    ref, _ = oracle.render_sphere_lights(lights, env, dfu, oracle.make_texture(np.ascontiguousarray(got), abi.GBUFFER_FLOAT4),
                                         oracle.make_texture(atlas, abi.SDF_UNORM16), (0.05, 0.05, 0.05, 1.0), w, h)
    assert_close(lit, ref, "lightmap from the generated 2.5D G-buffer")
    for x in (lm, sdf, gb):
        x.close()


def test_mesh_entry_point_validates_its_arguments(ctx):
    gb = native.GBufferTexture(ctx, None, abi.GBUFFER_FLOAT4, size=(32, 32))
    tri = scenes.top_face_mesh([(2, 2), (20, 2), (10, 20)], 0, 4)
    with pytest.raises(native.IlluminantError):
        gb.render_meshes(scenes.gbuffer_mesh_desc(two_point_five_d=False), tri, tri)              # front faces without 2.5D
    with pytest.raises(native.IlluminantError):
        gb.render_meshes(scenes.gbuffer_mesh_desc(extent_z=0.0), tri)                             # no depth range
    with pytest.raises(native.IlluminantError):
        gb.render_meshes(scenes.gbuffer_mesh_desc(), tri[:2])                                     # not a triangle list
    bb = scenes.billboard_vertices([dict(screen_bounds=((1, 1), (5, 5)))])
    with pytest.raises(native.IlluminantError):
        gb.render_meshes(scenes.gbuffer_mesh_desc(), billboards=bb, runs=[(None, 0, 2, abi.BILLBOARD_MASK)])    # run past the array
    with pytest.raises(native.IlluminantError):
        gb.render_meshes(scenes.gbuffer_mesh_desc(), billboards=bb, runs=[(None, 0, 1, 7)])
    gb.render_meshes(scenes.gbuffer_mesh_desc(), tri)
    gb.close()


def test_billboard_texture_bounds_outside_the_unit_square_are_clamped_not_clipped(ctx, oracle):
    """A billboard's depth is 0 (BillboardVertex.POSITION0 is a Vector2): its fragments are never clipped against the near / far plane,
    and TexCoord outside [0, 1] reads the clamped border texel (the closed form lives in tests/test_gbuffer_kat.py)."""
    w, h = 48, 40
    d = scenes.gbuffer_mesh_desc(two_point_five_d=True, z_to_y=1.0)
    tex = np.zeros((2, 2, 4), np.uint8)
    tex[..., 3] = [[255, 255], [255, 0]]
    bb = scenes.billboard_vertices([dict(screen_bounds=((8.0, 4.0), (24.0, 36.0)), texture_bounds=((-0.5, -0.5), (1.5, 1.5)))], 0.0, 1.0)
    lm = native.Lightmap(ctx, 2, 2, abi.LIGHTMAP_RGBA8)
    lm.upload(tex)
    gb = native.GBufferTexture(ctx, None, abi.GBUFFER_FLOAT4, size=(w, h))
    gb.render_meshes(d, None, None, bb, [(lm, 0, 1, abi.BILLBOARD_MASK)])
    got = gb.download()
    want = oracle.render_gbuffer_meshes(w, h, d, billboards=bb, runs=[(0, 1, abi.BILLBOARD_MASK)], textures=[tex])
    compare(got, want, abi.GBUFFER_FLOAT4)
    ground = np.float32([0.5, 1.0, 0.0, 1.0])
    drawn = ~np.all(got == ground, axis=-1)
    assert drawn[4, 8] and drawn[35, 8] and drawn[4, 23] and not drawn[35, 23] and drawn.sum() == 16 * 32 - 8 * 16      # the transparent texel covers u, v >= 0.5: columns 16-23, rows 20-35
    lm.close(); gb.close()


@pytest.mark.parametrize("scale", [(-1.0, 1.0), (1.0, -1.25), (-1.125, -1.0), (0.0, 1.0)])
def test_a_mirrored_view_is_refused_not_drawn_without_its_ground(ctx, scale):
    """under a negative ViewportScale the rasteriser's set-up reorders the ground quad's vertices, and the rectangle shortcut of
    the ground plane looked at fixed vertex slots.  The entry point refuses such a view outright (ViewportScale must be positive: the
    reference's view transforms scale by positive factors); the shortcut itself now takes the rectangle from the extremes of the snapped
    corners and falls back to the quad's two triangles when they do not form one (gbuffer.hip, gbuffer_setup_kernel)."""
    top = scenes.top_face_mesh([(-40.0, -30.0), (-10.0, -30.0), (-10.0, -5.0), (-40.0, -5.0)], 0.0, 12.0)
    d = scenes.gbuffer_mesh_desc(ground_z=2.0, viewport_scale=scale, z_to_y=0.0, extent_z=64.0, two_point_five_d=False)
    gb = native.GBufferTexture(ctx, None, abi.GBUFFER_FLOAT4, size=(96, 72))
    with pytest.raises(native.IlluminantError) as e:
        gb.render_meshes(d, top, None, None, [])
    assert e.value.code == abi.ERR_INVALID_ARGUMENT and "ViewportScale" in str(e.value)
    gb.close()
