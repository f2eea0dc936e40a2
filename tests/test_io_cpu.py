"""CPU tests of the on-disk formats either side of the hot path (neuma_amd/io.py).  Files are built
byte by byte / with json + torch.save exactly as the reference's writers lay them out, independently of our writers."""
import json
import math
import struct

import numpy as np
import pytest
import torch

from neuma_amd import io as nio

GOLD = __import__("pathlib").Path(__file__).resolve().parent / "golden"


def _handmade_kernels_ply(path, K, deg, rng):
    """plyfile layout of GaussianModel.save_ply (gaussian_model.py:189-220): float32 properties in construct_list_of_attributes order."""
    n_rest = 3 * (deg + 1) ** 2 - 3
    names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(n_rest)] + \
            ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
    vals = rng.normal(size=(K, len(names))).astype("<f4")
    vals[:, 3:6] = 0.0                                  # save_ply writes zero normals (gaussian_model.py:207)
    hdr = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % K + "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(hdr.encode("ascii"))
        for row in vals:
            f.write(struct.pack("<%df" % len(names), *row))
    return names, vals


@pytest.mark.parametrize("deg", [0, 3])
def test_kernels_ply_reader_follows_the_reference_property_layout(tmp_path, deg):
    rng = np.random.default_rng(0)
    K = 37
    names, vals = _handmade_kernels_ply(tmp_path / "kernels.ply", K, deg, rng)
    g = nio.load_gaussians_ply(tmp_path / "kernels.ply", deg)
    col = {n: vals[:, i] for i, n in enumerate(names)}
    assert np.array_equal(g.get_xyz.numpy(), np.stack([col["x"], col["y"], col["z"]], 1))
    assert g._features_dc.shape == (K, 1, 3) and g._features_rest.shape == (K, (deg + 1) ** 2 - 1, 3)
    # load_ply: features_dc[:, c, 0] = f_dc_c ; f_rest reshaped (K, 3, M-1) then transposed (gaussian_model.py:236-262)
    assert np.array_equal(g._features_dc[:, 0, :].numpy(), np.stack([col[f"f_dc_{c}"] for c in range(3)], 1))
    M1 = (deg + 1) ** 2 - 1
    for c in range(3):
        for m in range(M1):
            assert np.array_equal(g._features_rest[:, m, c].numpy(), col[f"f_rest_{c * M1 + m}"])
    assert np.array_equal(g._opacity[:, 0].numpy(), col["opacity"])
    assert np.array_equal(g._scaling.numpy(), np.stack([col[f"scale_{i}"] for i in range(3)], 1))
    assert np.array_equal(g._rotation.numpy(), np.stack([col[f"rot_{i}"] for i in range(4)], 1))
    assert g.get_features.shape == (K, (deg + 1) ** 2, 3) and g.active_sh_degree == deg
    # our writer produces the same bytes as the hand-made file, and the masked loader subsets rows
    nio.save_gaussians_ply(g, tmp_path / "again.ply")
    assert (tmp_path / "again.ply").read_bytes() == (tmp_path / "kernels.ply").read_bytes()
    mask = np.arange(K) % 3 == 0
    gm = nio.load_gaussians_ply(tmp_path / "kernels.ply", deg, mask=mask)
    assert np.array_equal(gm.get_xyz.numpy(), g.get_xyz.numpy()[mask])
    with pytest.raises(AssertionError):
        nio.load_gaussians_ply(tmp_path / "kernels.ply", 1 if deg != 1 else 2)


def test_particles_ply_variants(tmp_path):
    rng = np.random.default_rng(1)
    P = rng.random((11, 3))
    # trimesh-style binary point cloud with float vertices + uchar colours
    hdr = "ply\nformat binary_little_endian 1.0\ncomment https://github.com/mikedh/trimesh\nelement vertex 11\n" \
          "property float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n" \
          "property uchar alpha\nend_header\n"
    with open(tmp_path / "a.ply", "wb") as f:
        f.write(hdr.encode())
        for p in P:
            f.write(struct.pack("<3f4B", *p, 1, 2, 3, 255))
    assert np.allclose(nio.load_particles_ply(tmp_path / "a.ply"), P.astype(np.float32))
    # double-precision, big-endian
    hdr = "ply\nformat binary_big_endian 1.0\nelement vertex 11\nproperty double x\nproperty double y\nproperty double z\nend_header\n"
    with open(tmp_path / "b.ply", "wb") as f:
        f.write(hdr.encode())
        for p in P:
            f.write(struct.pack(">3d", *p))
    assert np.array_equal(nio.load_particles_ply(tmp_path / "b.ply"), P)
    # ascii with a trailing face element
    with open(tmp_path / "c.ply", "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 11\nproperty float x\nproperty float y\nproperty float z\n"
                "element face 0\nproperty list uchar int vertex_indices\nend_header\n")
        for p in P:
            f.write("%.9g %.9g %.9g\n" % tuple(p))
    assert np.allclose(nio.load_particles_ply(tmp_path / "c.ply"), P, atol=1e-6)
    nio.save_particles_ply(tmp_path / "d.ply", P)
    assert np.allclose(nio.load_particles_ply(tmp_path / "d.ply"), P.astype(np.float32))
    (tmp_path / "bad.ply").write_bytes((tmp_path / "a.ply").read_bytes()[:-5])
    with pytest.raises(ValueError):
        nio.load_particles_ply(tmp_path / "bad.ply")


def test_bindings_pt_to_csr(tmp_path):
    rng = np.random.default_rng(2)
    K, N = 13, 29
    dense = np.zeros((K, N), dtype=np.float32)
    for r in range(K):
        cols = rng.choice(N, size=rng.integers(0, 6), replace=False)
        dense[r, cols] = 1.0 / max(len(cols), 1)
    sp = torch.tensor(dense).to_sparse_coo()
    torch.save({"bindings_ind": sp.indices().cpu(), "bindings_val": sp.values().cpu(), "bindings_size": sp.size(),
                "n_particles": torch.tensor((dense > 0).sum(1))}, tmp_path / "bindings.pt")       # tune/utils.py:309-317
    b, n_p = nio.load_bindings(tmp_path / "bindings.pt")
    assert (b.K, b.N) == (K, N) and np.array_equal(n_p.numpy(), (dense > 0).sum(1).astype(np.float32))
    rowptr, col, val = b.rowptr.numpy(), b.col.numpy(), b.val.numpy()
    rebuilt = np.zeros_like(dense)
    for r in range(K):
        for q in range(rowptr[r], rowptr[r + 1]):
            rebuilt[r, col[q]] += val[q]
    assert np.array_equal(rebuilt, dense)
    rp_t, col_t, val_t = b.t_rowptr.numpy(), b.t_col.numpy(), b.t_val.numpy()
    rebuilt_t = np.zeros((N, K), dtype=np.float32)
    for r in range(N):
        for q in range(rp_t[r], rp_t[r + 1]):
            rebuilt_t[r, col_t[q]] += val_t[q]
    assert np.array_equal(rebuilt_t, dense.T)


def test_neuma_synthetic_camera_reader(tmp_path):
    root = tmp_path / "scene"
    (root / "data_dynamic").mkdir(parents=True)
    entries = []
    rng = np.random.default_rng(3)
    truth = {}
    for view in ("r_0", "r_1"):
        ang = rng.uniform(0, 2 * math.pi)
        Rm = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
        c2w = np.concatenate([Rm, rng.normal(size=(3, 1))], 1)                 # (3,4) form
        for step in (0, 1, 2):
            name = f"./data_dynamic/{view}_{step:03d}.png"
            (root / name).write_bytes(b"")
            entries.append({"file_path": name, "c2w": c2w.tolist(), "intrinsic": [[400.0, 0, 128.0], [0, 420.0, 96.0], [0, 0, 1]]})
            truth[(view, step)] = c2w
    (root / "data_dynamic.json").write_text(json.dumps(entries))
    out = nio.read_neuma_synthetic_cameras(str(root), "data_dynamic.json", True, exclude_steps=[2])
    assert out["views"] == ["r_0", "r_1"] and out["steps"] == [0, 1]
    assert [(c.view, c.step) for c in out["cam_infos"]] == [("r_0", 0), ("r_0", 1), ("r_1", 0), ("r_1", 1)]
    for c in out["cam_infos"]:
        c2w = np.concatenate([truth[(c.view, c.step)], [[0, 0, 0, 1.0]]], 0)
        c2w[:3, 1:3] *= -1                                                     # dataset_readers.py:243-249
        w2c = np.linalg.inv(c2w)
        assert np.allclose(c.R, w2c[:3, :3].T) and np.allclose(c.T, w2c[:3, 3])
        assert (c.width, c.height) == (256, 192)
        assert math.isclose(c.FovX, 2 * math.atan(256 / 800.0)) and math.isclose(c.FovY, 2 * math.atan(192 / 840.0))
    only = nio.read_neuma_synthetic_cameras(str(root), "data_dynamic.json", True, init_frame=1, used_views=["r_1"])
    assert [(c.view, c.step) for c in only["cam_infos"]] == [("r_1", 1)]


def test_realcapture_camera_reader_and_rodrigues(tmp_path):
    from scipy.spatial.transform import Rotation
    for rv in ([0.3, -0.2, 0.9], [0.0, 0.0, 0.0], [1e-9, 0, 0], [2.0, 1.0, -2.5]):
        assert np.allclose(nio.rodrigues(rv), Rotation.from_rotvec(rv).as_matrix(), atol=1e-12)
    root = tmp_path / "cap"
    (root / "sparse/0").mkdir(parents=True)
    (root / "dynamics").mkdir()
    with open(root / "sparse/0/cameras.bin", "wb") as f:                        # COLMAP: PINHOLE (model 1) fx fy cx cy
        f.write(struct.pack("<Q", 1))
        f.write(struct.pack("<iiQQ", 1, 1, 4752, 2672))
        f.write(struct.pack("<4d", 3500.0, 3600.0, 2376.0, 1336.0))
    calib = {"cam_a": {"rvecs": [[0.1], [0.2], [0.3]], "tvecs": [[1.0], [2.0], [3.0]]}}
    (root / "cameras_calib.json").write_text(json.dumps(calib))
    for s in (0, 5):
        (root / "dynamics" / f"cam_a_{s}.jpg").write_bytes(b"")
    out = nio.read_realcapture_cameras(str(root), False)
    assert out["views"] == ["cam_a"] and out["steps"] == [0, 5] and len(out["cam_infos"]) == 2
    c = out["cam_infos"][1]
    assert np.allclose(c.R, Rotation.from_rotvec([0.1, 0.2, 0.3]).as_matrix().T) and np.allclose(c.T, [1, 2, 3])
    fx, fy = 3500.0 * 1920 / 4752, 3600.0 * 1080 / 2672
    assert math.isclose(c.FovY, 2 * math.atan(1080 / (2 * fx))) and math.isclose(c.FovX, 2 * math.atan(1920 / (2 * fy)))
    assert c.image_path.endswith("dynamics/cam_a_5.jpg") and (c.width, c.height) == (1920, 1080)


def test_disk_camera_matches_reference_matrices():
    """cameras.py:54-57 + graphics_utils.py:38-71, pinned by the golden produced with the reference's own modules."""
    z = np.load(GOLD / "camera_sh_golden.npz")
    info = nio.CameraInfo(0, z["R"], z["T"], float(z["fovy"]), float(z["fovx"]), "", 800, 600, "v", 0)
    cam = nio.DiskCamera(info)
    assert np.allclose(cam.world_view_transform.numpy(), z["world_view"], atol=1e-6)
    assert np.allclose(cam.projection_matrix.numpy(), z["proj"], atol=1e-6)
    assert np.allclose(cam.full_proj_transform.numpy(), z["full_proj"], atol=1e-5)
    assert np.allclose(cam.camera_center.numpy(), z["center"], atol=1e-5)


def test_init_and_lora_checkpoint_formats(tmp_path):
    torch.save({"init_x": torch.rand(5, 3), "init_v": torch.rand(5, 3)}, tmp_path / "init.pt")     # neuma_dataset.py:115-118
    x, v = nio.load_init_state(tmp_path / "init.pt")
    assert x.shape == (5, 3) and v.shape == (5, 3)


def test_mesh_sampling_and_init_frame_steps(tmp_path):
    """extras.mesh_sampling.sample_mesh_points (prepare.py's particle_data.mesh_path; outside the hot-path scope) on a closed box, and the camera readers' `steps` entry with
    an init_frame (dataset_readers.py:230, 329: only that frame)."""
    import json
    import numpy as np
    from neuma_amd import io as nio
    v = np.array([[0, 0, 0], [2, 0, 0], [2, 1, 0], [0, 1, 0], [0, 0, 1], [2, 0, 1], [2, 1, 1], [0, 1, 1]], float)
    t = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [2, 3, 7], [2, 7, 6], [1, 2, 6], [1, 6, 5], [3, 0, 4], [3, 4, 7]])
    p = np.array([[1.0, 0.5, 0.5], [2.5, 0.5, 0.5], [0.2, 0.9, 0.1], [1.0, 0.5, -0.1], [1.0, 0.5, 1.1]])
    from neuma_amd.extras import mesh_sampling as mesh
    assert mesh.points_in_mesh(p, v, t).tolist() == [True, False, True, False, False]
    grid = mesh.sample_mesh_points(v, t, "volumetric", 10)
    assert len(grid) == 10 * 5 * 5 and grid.min() > 0 and (grid.max(0) < [2, 1, 1]).all()
    rnd = mesh.sample_mesh_points(v, t, "uniform", 8)
    assert len(rnd) == 8 ** 3                                        # the box fills its own bounding box
    # NeuMA-Synthetic layout: frames 0, 3, 7 of one view; init_frame = 3 -> steps == [3] (what dataset.steps / evaluate rely on)
    root = tmp_path / "scene"
    (root / "data_dynamic").mkdir(parents=True)
    entries = []
    for step in (0, 3, 7):
        name = f"./data_dynamic/r_0_{step:03d}.png"
        (root / name).write_bytes(b"")
        entries.append({"file_path": name, "c2w": np.eye(4)[:3].tolist(), "intrinsic": [[100.0, 0, 32.0], [0, 100.0, 24.0], [0, 0, 1]]})
    (root / "data_dynamic.json").write_text(json.dumps(entries))
    out = nio.read_neuma_synthetic_cameras(str(root), "data_dynamic.json", True, init_frame=3)
    assert out["steps"] == [3] and [c.step for c in out["cam_infos"]] == [3]
    assert nio.read_neuma_synthetic_cameras(str(root), "data_dynamic.json", True)["steps"] == [0, 3, 7]
