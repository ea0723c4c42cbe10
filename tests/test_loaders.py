"""Dataset side of the path: load_blender / load_llff / utils.load_datasets
against golden vectors captured from the reference's pose arithmetic (tests/golden/g9_loaders.npz,
made by tests/golden/make_golden_loaders.py) and against synthetic scenes written to disk in the two
formats.  CPU only."""
import json
import os
import sys
import types

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("NERF_AMD_QUIET", "1")

from nerf_shared_amd import image_io, load_blender, load_llff, utils  # noqa: E402

G = np.load(os.path.join(REPO, "tests", "golden", "g9_loaders.npz"))


def same(a, b, tol=0.0):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert a.dtype == b.dtype, (a.dtype, b.dtype)
    if tol == 0.0:
        assert np.array_equal(a, b)
    else:
        assert np.allclose(a, b, rtol=0, atol=tol), float(np.abs(a - b).max())


def test_pose_arithmetic_matches_reference_bit_for_bit():
    for args, want in zip(G["ps_args"], G["ps_out"]):
        same(load_blender.pose_spherical(*args).numpy(), want)
    v = G["vm_in"]
    same(load_llff.normalize(v[0]), G["normalize_out"])
    same(load_llff.viewmatrix(v[0], v[1], v[2]), G["viewmatrix_out"])
    # numpy's float32 reductions depend on the memory layout: rebuild p the way the fixture script (and
    # load_llff_data itself) does, from the poses_bounds layout, rather than from the C-ordered copy
    raw = G["raw_poses"]
    p = np.moveaxis(np.concatenate([raw[:, 1:2, :], -raw[:, 0:1, :], raw[:, 2:, :]], 1), -1, 0).astype(np.float32)
    assert np.array_equal(p, G["p_in"])
    same(load_llff.poses_avg(p), G["poses_avg_out"])
    same(load_llff.recenter_poses(p), G["recenter_out"])
    same(load_llff.ptstocam(G["ptstocam_pts"], p[0]), G["ptstocam_out"])
    c2w = load_llff.poses_avg(p)
    spiral = load_llff.render_path_spiral(c2w, load_llff.normalize(p[:, :3, 1].sum(0)), np.array([0.3, 0.2, 0.1]),
                                          3.5, 0.2, zrate=.5, rots=2, N=7)
    same(np.array(spiral), G["spiral_out"])
    bds = np.moveaxis(G["raw_bds"], -1, 0).astype(np.float32).copy()
    sp, ring, sb = load_llff.spherify_poses(p.copy(), bds)
    same(sp, G["spherify_poses"]); same(ring, G["spherify_render"]); same(sb, G["spherify_bds"])
    assert sb is bds                                           # scaled in place, like the reference


def write_llff_scene(root, factor):
    """poses_bounds.npy + images/ + images_<factor>/ holding the fixture's frames as PNGs."""
    poses, bds, imgs = G["raw_poses"], G["raw_bds"], G["raw_imgs"]
    n = poses.shape[-1]
    arr = np.concatenate([poses.transpose([2, 0, 1]).reshape(n, 15), bds.transpose([1, 0])], 1)
    np.save(os.path.join(root, "poses_bounds.npy"), arr)
    frames8 = (imgs * 255).astype(np.uint8)
    for d, scale in (("images", factor), ("images_%d" % factor, 1)):
        os.makedirs(os.path.join(root, d))
        for i in range(n):
            f = frames8[..., i]
            image_io.write_png(os.path.join(root, d, "%03d.png" % i), np.kron(f, np.ones((scale, scale, 1), np.uint8)))
    return frames8


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_load_llff_data_matches_reference(case, tmp_path, monkeypatch):
    recenter, bd_factor, spherify, path_zflat = G["llff%d_args" % case]
    kw = dict(recenter=bool(recenter), bd_factor=None if bd_factor < 0 else float(bd_factor),
              spherify=bool(spherify), path_zflat=bool(path_zflat))
    # (a) the arithmetic after the read, on exactly the arrays the reference was given
    monkeypatch.setattr(load_llff, "_load_data", lambda basedir, factor=None, **_: (
        G["raw_poses"].copy(), G["raw_bds"].copy(), G["raw_imgs"].copy()))
    images, poses, bds, render_poses, i_test = load_llff.load_llff_data("unused", factor=8, **kw)
    same(images, G["llff%d_images" % case]); same(poses, G["llff%d_poses" % case]); same(bds, G["llff%d_bds" % case])
    same(render_poses, G["llff%d_render_poses" % case])
    assert int(i_test) == int(G["llff%d_i_test" % case])
    monkeypatch.undo()
    # (b) the same scene through the files: only the 8-bit quantisation of the frames and the
    #     H, W, focal/factor column written by _load_data differ
    frames8 = write_llff_scene(str(tmp_path), 4)
    images, poses, bds, render_poses, i_test = load_llff.load_llff_data(str(tmp_path), factor=4, **kw)
    same(images, np.moveaxis(frames8 / 255., -1, 0).astype(np.float32))
    same(poses[:, :, :4], G["llff%d_poses" % case][:, :, :4])
    assert np.array_equal(poses[0, :, 4], np.array([12, 16, 11.5 / 4], np.float32))
    same(bds, G["llff%d_bds" % case])
    assert int(i_test) == int(G["llff%d_i_test" % case])


def test_llff_path_zflat_and_minify(tmp_path):
    """path_zflat=True (the reference cannot run it under numpy >= 1.18): 60 poses on one flat turn;
    a missing images_<factor> folder is produced from images/."""
    write_llff_scene(str(tmp_path), 2)
    out = load_llff.load_llff_data(str(tmp_path), factor=2, path_zflat=True)
    assert out[3].shape == (60, 3, 5) and out[3].dtype == np.float32
    import shutil
    shutil.rmtree(tmp_path / "images_2")
    pytest.importorskip("PIL")
    images = load_llff.load_llff_data(str(tmp_path), factor=2)[0]
    assert images.shape == (9, 12, 16, 3) and os.path.isdir(tmp_path / "images_2")
    assert np.abs(images - np.moveaxis(G["raw_imgs"], -1, 0)).mean() < 0.12      # resampled noise: close, not identical


def write_blender_scene(root, H, W, with_bounds):
    rng = np.random.default_rng(1)
    frames = {}
    k = 0
    for split, n in (("train", 5), ("val", 3), ("test", 9)):
        meta = {"camera_angle_x": 0.6911112070083618, "frames": []}
        if with_bounds:
            meta["near"], meta["far"] = 1.5, 7.25
        os.makedirs(os.path.join(root, split), exist_ok=True)
        for i in range(n):
            img = rng.integers(0, 256, size=(H, W, 4), dtype=np.uint8)
            image_io.write_png(os.path.join(root, split, "r_%d.png" % i), img)
            pose = np.eye(4); pose[:3, 3] = rng.normal(size=3); pose[0, 1] = 0.001 * k
            meta["frames"].append({"file_path": "./%s/r_%d" % (split, i), "transform_matrix": pose.tolist()})
            frames[(split, i)] = (img, pose)
            k += 1
        with open(os.path.join(root, "transforms_%s.json" % split), "w") as f:
            json.dump(meta, f)
    return frames


@pytest.mark.parametrize("half_res", [False, True])
def test_load_blender_data(tmp_path, half_res):
    H, W = 8, 12
    frames = write_blender_scene(str(tmp_path), H, W, with_bounds=True)
    imgs, poses, render_poses, hwf, i_split, near, far = load_blender.load_blender_data(str(tmp_path), half_res, testskip=4)
    # train: all 5; val / test: every 4th
    assert [len(s) for s in i_split] == [5, 1, 3] and imgs.shape[0] == 9
    assert (near, far) == (1.5, 7.25)
    focal = .5 * W / np.tan(.5 * 0.6911112070083618)
    order = [("train", i) for i in range(5)] + [("val", 0)] + [("test", i) for i in (0, 4, 8)]
    same(poses, np.array([frames[k][1] for k in order]).astype(np.float32))
    full = (np.array([frames[k][0] for k in order]) / 255.).astype(np.float32)
    if not half_res:
        assert hwf == [H, W, focal]
        same(imgs, full)
    else:
        assert hwf == [H // 2, W // 2, focal / 2.]
        assert imgs.dtype == np.float64 and imgs.shape == (9, H // 2, W // 2, 4)
        want = full.reshape(9, H // 2, 2, W // 2, 2, 4).astype(np.float64).mean((2, 4))
        assert np.abs(imgs - want).max() < 1e-6
        assert np.array_equal(imgs, imgs.astype(np.float32))          # fp32 values in a float64 container
    assert render_poses.shape == (40, 4, 4)
    same(render_poses[0].numpy(), G["ps_out"][0]); same(render_poses[1].numpy(), G["ps_out"][1])


def test_halve_area_hand_computed_fixture():
    """half_res (load_blender.py:86-94) resizes with cv2.INTER_AREA, which for an exact factor of 2 is the mean
    of every 2x2 block.  cv2 is absent offline, so this stage is UNPINNED against cv2 itself; what is pinned
    is the arithmetic, on a hand-computed fixture whose block means are exact in fp32."""
    img = (np.arange(16, dtype=np.float32).reshape(4, 4) / 16.0)[..., None] * np.array([1.0, 2.0, 0.5, 1.0], np.float32)
    # rows 0-1: blocks {0,1,4,5} -> 2.5 and {2,3,6,7} -> 4.5; rows 2-3: {8,9,12,13} -> 10.5 and {10,11,14,15} -> 12.5
    want = (np.array([[2.5, 4.5], [10.5, 12.5]], np.float32) / 16.0)[..., None] * np.array([1.0, 2.0, 0.5, 1.0], np.float32)
    got = load_blender.halve_area(img)
    assert got.dtype == np.float32 and got.shape == (2, 2, 4)
    assert np.array_equal(got, want)
    # 8-bit pixels over 255 (what the loader feeds it): within one fp32 ulp of the exact rational mean
    px = np.array([[[255, 0, 17, 255], [254, 1, 18, 0]], [[3, 2, 19, 255], [0, 255, 20, 1]]], np.uint8)
    got = load_blender.halve_area((px / 255.).astype(np.float32))
    exact = px.astype(np.float64).sum((0, 1)) / (4 * 255.0)                 # [512, 258, 74, 511] / 1020
    assert np.abs(got[0, 0].astype(np.float64) - exact).max() <= np.spacing(np.float32(0.5))
    # an odd edge is cropped (stock NeRF-synthetic frames are 800x800; cv2 would use a fractional footprint there)
    assert load_blender.halve_area(np.ones((5, 7, 4), np.float32)).shape == (2, 3, 4)


def test_load_datasets_blender_and_llff(tmp_path):
    b, l = tmp_path / "b", tmp_path / "l"
    os.makedirs(b); os.makedirs(l)
    write_blender_scene(str(b), 8, 12, with_bounds=False)          # stock JSON: no near / far keys
    args = types.SimpleNamespace(dataset_type="blender", datadir=str(b), half_res=False, testskip=1, white_bkgd=True,
                                 render_test=False)
    images, poses, render_poses, hwf, i_split, K, bd = utils.load_datasets(args)
    assert images.shape == (17, 8, 12, 3) and bd == {"near": 2.0, "far": 6.0}
    assert K.shape == (3, 3) and K[0, 2] == 6.0 and K[1, 2] == 4.0 and K[0, 0] == hwf[2]
    raw = load_blender.load_blender_data(str(b), False, 1)[0]
    same(images, raw[..., :3] * raw[..., -1:] + (1. - raw[..., -1:]))
    write_llff_scene(str(l), 4)
    args = types.SimpleNamespace(dataset_type="llff", datadir=str(l), factor=4, spherify=False, llffhold=4, no_ndc=False,
                                 render_test=True)
    images, poses, render_poses, hwf, (i_train, i_val, i_test), K, bd = utils.load_datasets(args)
    assert poses.shape == (9, 3, 4) and hwf[:2] == [12, 16] and bd == {"near": 0., "far": 1.}
    assert list(i_test) == [0, 4, 8] and list(i_train) == [1, 2, 3, 5, 6, 7]
    same(render_poses, poses[i_test])
    args.no_ndc = True
    bd2 = utils.load_datasets(args)[-1]
    assert bd2["near"] > 0 and bd2["far"] > bd2["near"]
    with pytest.raises(NotImplementedError):
        utils.load_datasets(types.SimpleNamespace(dataset_type="deepvoxels"))
