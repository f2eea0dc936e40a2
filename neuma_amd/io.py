"""On-disk formats either side of the hot path: what the reference's drivers read before they reach
the kernels and write after them.  Host-side only (numpy / json / torch.load), no third-party readers:

  kernels.ply      3DGS Gaussians, binary little-endian PLY, one `vertex` element with float32 properties in the order
                   x,y,z,nx,ny,nz,f_dc_*,f_rest_*,opacity,scale_*,rot_*   (gaussian_model.py:189-201 construct_list_of_attributes,
                   203-220 save_ply, 227-270 load_ply; `plyfile` is not available, so this is an own parser/writer)
  particles.ply    simulation particles as a PLY point cloud (tune/utils.py:304-307 writes it through trimesh.PointCloud.export;
                   inference.py:192 / neuma_instance.py:165 read `.vertices`)
  bindings.pt      torch.save({'bindings_ind' (2,nnz) int64, 'bindings_val' (nnz,), 'bindings_size' (K,N), 'n_particles' (K,)})
                   (tune/utils.py:309-317; consumed at render.py:202-207)
  init.pt          {'init_x', 'init_v'} (render.py:222-223, neuma_dataset.py:115-118)
  *_lora.pt        {'elasticity', 'plasticity', 'loss'} LoRA state dicts (finetune.py:470-476) - see train.py
  data_dynamic.json / transforms  NeuMA-Synthetic cameras: per image `c2w` (3x4 or 4x4, OpenGL axes) + `intrinsic` (3x3)
                   (dataset_readers.py:200-279)
  cameras_calib.json + sparse/0/cameras.bin   RealCapture cameras: Rodrigues rvecs / tvecs per view + COLMAP intrinsics
                   (dataset_readers.py:282-372)
"""
import json
import math
import os
import struct
from pathlib import Path
from typing import Dict, List, NamedTuple, Optional, Sequence, Tuple

import numpy as np
import torch

# ------------------------------------------------------------------ PLY (vertex element only)

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}


def _ply_header(f, path):
    """Parse a PLY header from the open binary file `f` (positioned at the start); returns (format, elements) with
    elements = [{'name', 'count', 'props': [(dtype | 'list', name, (count type, item type) | None)]}] and leaves `f` at the data."""
    if f.readline().strip() != b"ply":
        raise ValueError(f"{path}: not a PLY file")
    fmt, elements, cur = None, [], None
    while True:
        line = f.readline()
        if not line:
            raise ValueError(f"{path}: unterminated PLY header")
        tok = line.decode("ascii", "replace").split()
        if not tok or tok[0] in ("comment", "obj_info"):
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            cur = {"name": tok[1], "count": int(tok[2]), "props": []}
            elements.append(cur)
        elif tok[0] == "property":
            if tok[1] == "list":
                cur["props"].append(("list", tok[-1], (tok[2], tok[3])))
            else:
                if tok[1] not in _PLY_TYPES:
                    raise ValueError(f"{path}: unknown PLY type {tok[1]}")
                cur["props"].append((_PLY_TYPES[tok[1]], tok[2], None))
        elif tok[0] == "end_header":
            break
    if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
        raise ValueError(f"{path}: unsupported PLY format {fmt}")
    if not elements or elements[0]["name"] != "vertex":
        raise ValueError(f"{path}: first element must be `vertex`")
    if any(p[0] == "list" for p in elements[0]["props"]):
        raise ValueError(f"{path}: list properties in the vertex element are not supported")
    return fmt, elements


def _ply_vertex_block(f, path, fmt, el) -> Dict[str, np.ndarray]:
    n = el["count"]
    if fmt == "ascii":
        rows = [f.readline().split() for _ in range(n)]
        cols = list(zip(*rows)) if n else [[] for _ in el["props"]]
        return {name: np.asarray(col, dtype=np.float64).astype(t) for (t, name, _), col in zip(el["props"], cols)}
    end = "<" if fmt == "binary_little_endian" else ">"
    dt = np.dtype([(name, end + t) for t, name, _ in el["props"]])
    raw = f.read(dt.itemsize * n)
    if len(raw) != dt.itemsize * n:
        raise ValueError(f"{path}: truncated vertex data ({len(raw)} of {dt.itemsize * n} bytes)")
    arr = np.frombuffer(raw, dtype=dt, count=n)
    return {name: np.ascontiguousarray(arr[name]).astype(arr[name].dtype.newbyteorder("=")) for _, name, _ in el["props"]}


def read_ply_vertices(path) -> Dict[str, np.ndarray]:
    """Properties of the `vertex` element of a PLY file (ascii, binary_little_endian or binary_big_endian) as a dict of
    1-D arrays, in file order.  Other elements (faces ...) are ignored; list properties inside `vertex` are rejected."""
    with open(path, "rb") as f:
        fmt, elements = _ply_header(f, path)
        return _ply_vertex_block(f, path, fmt, elements[0])


def write_ply_vertices(path, names: Sequence[str], data: np.ndarray) -> None:
    """Binary little-endian PLY with one `vertex` element of float32 properties `names` (columns of `data`), the layout
    plyfile's PlyData([PlyElement.describe(elements, 'vertex')]).write() produces for an all-'f4' structured array."""
    data = np.ascontiguousarray(np.asarray(data, dtype="<f4"))
    assert data.ndim == 2 and data.shape[1] == len(names)
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {data.shape[0]}"]
    header += [f"property float {n}" for n in names] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(data.tobytes())


# ------------------------------------------------------------------ kernels.ply  <->  GaussianModel

def gaussian_attribute_names(n_dc: int, n_rest: int, n_scale: int = 3, n_rot: int = 4) -> List[str]:
    """gaussian_model.py:189-201"""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)] + [f"f_rest_{i}" for i in range(n_rest)]
    names += ["opacity"] + [f"scale_{i}" for i in range(n_scale)] + [f"rot_{i}" for i in range(n_rot)]
    return names


def save_gaussians_ply(gaussians, path) -> None:
    """gaussian_model.py:203-220: features are stored channel-major ((K,C,3) -> transpose(1,2) -> flatten)."""
    xyz = gaussians._xyz.detach().cpu().numpy()
    f_dc = gaussians._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    f_rest = gaussians._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    opac = gaussians._opacity.detach().cpu().numpy()
    scale = gaussians._scaling.detach().cpu().numpy()
    rot = gaussians._rotation.detach().cpu().numpy()
    attrs = np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, opac, scale, rot), axis=1)
    write_ply_vertices(path, gaussian_attribute_names(f_dc.shape[1], f_rest.shape[1], scale.shape[1], rot.shape[1]), attrs)


def load_gaussians_ply(path, sh_degree: int, device="cpu", mask: Optional[np.ndarray] = None):
    """gaussian_model.py:227-270 (load_ply) / 272-318 (load_ply_with_mask): returns a GaussianModel with
    active_sh_degree = max_sh_degree = sh_degree.  f_rest_* / scale_* / rot_* are ordered by their numeric suffix."""
    from .render.gaussian_model import GaussianModel
    v = read_ply_vertices(path)
    xyz = np.stack((v["x"], v["y"], v["z"]), axis=1)
    opac = np.asarray(v["opacity"])[..., None]
    K = xyz.shape[0]
    fdc = np.zeros((K, 3, 1))
    for c in range(3):
        fdc[:, c, 0] = v[f"f_dc_{c}"]

    def family(prefix):
        names = sorted([n for n in v if n.startswith(prefix)], key=lambda s: int(s.split("_")[-1]))
        return np.stack([v[n] for n in names], axis=1) if names else np.zeros((K, 0))

    rest = family("f_rest_")
    assert rest.shape[1] == 3 * (sh_degree + 1) ** 2 - 3, \
        f"{path}: {rest.shape[1]} f_rest_* properties do not match sh_degree={sh_degree}"       # gaussian_model.py:241
    rest = rest.reshape(K, 3, (sh_degree + 1) ** 2 - 1)
    scales, rots = family("scale_"), family("rot")
    if mask is not None:
        xyz, opac, fdc, rest, scales, rots = (a[mask] for a in (xyz, opac, fdc, rest, scales, rots))
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)
    gm = GaussianModel(sh_degree)
    gm.set_params(t(xyz), t(fdc).transpose(1, 2).contiguous(), t(rest).transpose(1, 2).contiguous(), t(scales), t(rots), t(opac))
    return gm


# ------------------------------------------------------------------ particles.ply, bindings.pt, init.pt

def load_particles_ply(path) -> np.ndarray:
    """(N,3) float64 vertices of a PLY point cloud (trimesh.load(path).vertices in the reference)."""
    v = read_ply_vertices(path)
    return np.stack((v["x"], v["y"], v["z"]), axis=1).astype(np.float64)


def save_particles_ply(path, points) -> None:
    write_ply_vertices(path, ["x", "y", "z"], np.asarray(points, dtype=np.float32).reshape(-1, 3))


def load_bindings(path, device=None):
    """bindings.pt -> (tune.Bindings with CSR + transposed CSR on `device`, n_particles (K,) float32).
    render.py:202-207 rebuilds a sparse COO tensor from the same four entries."""
    from .tune import Bindings
    d = torch.load(path, map_location="cpu")
    ind, val, size = d["bindings_ind"], d["bindings_val"], tuple(int(s) for s in d["bindings_size"])
    return Bindings(ind.long(), val.float(), size, device), d["n_particles"].float().to(device if device is not None else "cpu")


def save_bindings(path, indices: torch.Tensor, values: torch.Tensor, size, n_particles: torch.Tensor) -> None:
    """tune/utils.py:309-317"""
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    torch.save({"bindings_ind": indices.cpu(), "bindings_val": values.cpu(), "bindings_size": torch.Size(size),
                "n_particles": n_particles.cpu()}, path)


def load_init_state(path) -> Tuple[torch.Tensor, torch.Tensor]:
    """init.pt of stage A (optimize_init_velocity): render.py:222-223"""
    d = torch.load(path, map_location="cpu")
    return d["init_x"], d["init_v"]


# ------------------------------------------------------------------ cameras

class CameraInfo(NamedTuple):
    """PhysCameraInfo of dataset_readers.py (image optional: only loaded on request)."""
    uid: int
    R: np.ndarray          # (3,3), stored transposed ("glm" convention, dataset_readers.py:249)
    T: np.ndarray          # (3,)
    FovY: float
    FovX: float
    image_path: str
    width: int
    height: int
    view: str
    step: int
    image: Optional[np.ndarray] = None     # (H,W,3) float32 in [0,1], alpha-composited on the background


def focal2fov(focal: float, pixels: float) -> float:
    """graphics_utils.py:76-77"""
    return 2 * math.atan(pixels / (2 * focal))


def fov2focal(fov: float, pixels: float) -> float:
    return pixels / (2 * math.tan(fov / 2))


def _views_and_steps(folder, exclude_steps, used_views):
    views, steps = set(), set()
    for d in os.listdir(folder):
        stem, step = d.rsplit("_", 1)
        if used_views is None or stem in used_views:
            views.add(str(stem))
        s = int(step.split(".")[0])
        if s not in exclude_steps:
            steps.add(s)
    return sorted(views), sorted(steps)


def _load_image(path, white_background: bool) -> np.ndarray:
    from PIL import Image
    im = np.array(Image.open(path).convert("RGBA")) / 255.0
    bg = np.ones(3) if white_background else np.zeros(3)
    return (im[:, :, :3] * im[:, :, 3:4] + bg * (1 - im[:, :, 3:4])).astype(np.float32)


def read_neuma_synthetic_cameras(path, transformsfile: str, white_background: bool, extension: str = ".png", init_frame=None,
                                 exclude_steps=(-1,), used_views=None, load_images: bool = False,
                                 image_size: Optional[Tuple[int, int]] = None) -> Dict:
    """dataset_readers.py:200-279.  `c2w` is OpenGL/Blender (Y up, Z back): columns 1,2 are negated to reach COLMAP axes,
    R = (w2c[:3,:3])^T, T = w2c[:3,3]; FoV from the intrinsic's focal lengths and the image size.  Without load_images the
    size comes from `image_size` (W,H) or from the intrinsic's principal point (2*cx, 2*cy)."""
    subfolder = transformsfile.split(".")[0]
    views, steps = _views_and_steps(os.path.join(path, subfolder), set(exclude_steps), used_views)
    with open(os.path.join(path, transformsfile)) as f:
        contents = json.load(f)
    meta = {}
    for entry in contents:
        e = dict(entry)
        meta[e.pop("file_path")] = e
    steps_used = [init_frame] if init_frame is not None else steps
    infos, idx = [], 0
    for view in views:
        for step in steps_used:
            key = f"./{subfolder}/{view}_{step:03d}{extension}"
            assert key in meta, f"File {key} not found in meta_info!"
            c2w = np.array(meta[key]["c2w"], dtype=np.float64)
            if c2w.shape[0] == 3:
                c2w = np.concatenate([c2w, np.array([[0, 0, 0, 1.0]])], axis=0)
            c2w[:3, 1:3] *= -1
            w2c = np.linalg.inv(c2w)
            R, T = np.transpose(w2c[:3, :3]), w2c[:3, 3]
            K = np.array(meta[key]["intrinsic"], dtype=np.float64)
            img_path = os.path.join(path, key)
            image = _load_image(img_path, white_background) if load_images else None
            if image is not None:
                H, W = image.shape[:2]
            elif image_size is not None:
                W, H = image_size
            else:
                W, H = int(round(2 * K[0][2])), int(round(2 * K[1][2]))
            infos.append(CameraInfo(idx, R, T, focal2fov(K[1][1], H), focal2fov(K[0][0], W), img_path, W, H, view, step, image))
            idx += 1
    return {"cam_infos": infos, "views": views, "steps": steps_used}      # (dataset_readers.py:230: [init_frame] when one is given)


def rodrigues(rvec) -> np.ndarray:
    """Rotation vector -> matrix (cv2.Rodrigues)."""
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * Kx + (1 - math.cos(th)) * (Kx @ Kx)


_COLMAP_NPARAMS = {0: 3, 1: 4, 2: 4, 3: 5, 4: 8, 5: 8, 6: 12, 7: 5, 8: 4, 9: 5, 10: 12}


def read_colmap_cameras_bin(path) -> Dict[int, Dict]:
    """colmap_loader.py read_intrinsics_binary: {camera_id: {model_id, width, height, params}}."""
    out = {}
    with open(path, "rb") as f:
        n = struct.unpack("<Q", f.read(8))[0]
        for _ in range(n):
            cam_id, model_id, w, h = struct.unpack("<iiQQ", f.read(24))
            k = _COLMAP_NPARAMS[model_id]
            params = np.array(struct.unpack("<" + "d" * k, f.read(8 * k)))
            out[cam_id] = {"model_id": model_id, "width": w, "height": h, "params": params}
    return out


def _load_masked_image(image_path: str, extension: str, white_background: bool, read_mask_only: bool) -> np.ndarray:
    """dataset_readers.py:338-352: RealCapture frames come with a binary mask under dynamic_masks/<same name>.png; either the
    mask itself is the image (read_mask_only: 3 equal channels) or the photo is composited onto the background through it."""
    from PIL import Image
    mask_path = image_path.replace("/dynamics/", "/dynamic_masks/").replace(extension, ".png")
    mask = np.array(Image.open(mask_path))
    if mask.ndim == 3:
        mask = mask[:, :, 0]
    if read_mask_only:
        return (np.repeat(mask[:, :, None], 3, axis=-1) / 255.0).astype(np.float32)
    im = np.array(Image.open(image_path).convert("RGB")) / 255.0
    m = mask[:, :, None] / 255.0
    bg = np.ones(3) if white_background else np.zeros(3)
    arr = np.array((im * m + bg * (1 - m)) * 255.0, dtype=np.uint8)           # the reference quantises back to 8 bit here
    return (arr / 255.0).astype(np.float32)


def read_realcapture_cameras(path, white_background: bool, extension: str = ".jpg", width: int = 1920, height: int = 1080,
                             init_frame=None, exclude_steps=(-1,), used_views=None, load_images: bool = False,
                             read_mask_only: bool = False) -> Dict:
    """dataset_readers.py:282-372.  Intrinsics come from COLMAP camera 1, rescaled from the 4752x2672 capture size; NB the
    reference assigns FovY from fx/height and FovX from fy/width (:309-310) - kept."""
    intr = read_colmap_cameras_bin(os.path.join(path, "sparse/0", "cameras.bin"))
    fx = intr[1]["params"][0] * width / 4752
    fy = intr[1]["params"][1] * height / 2672
    FovY, FovX = focal2fov(fx, height), focal2fov(fy, width)
    with open(os.path.join(path, "cameras_calib.json")) as f:
        calib = json.load(f)
    views, steps = _views_and_steps(os.path.join(path, "dynamics"), set(exclude_steps), used_views)
    steps_used = [init_frame] if init_frame is not None else steps
    infos, idx = [], 0
    for view in views:
        R = np.transpose(rodrigues(calib[view]["rvecs"]))
        T = np.array(calib[view]["tvecs"], dtype=np.float64).reshape(3)
        for step in steps_used:
            img_path = os.path.join(path, f"./dynamics/{view}_{step}{extension}")
            image = None
            if load_images:
                has_mask = os.path.exists(img_path.replace("/dynamics/", "/dynamic_masks/").replace(extension, ".png"))
                image = _load_masked_image(img_path, extension, white_background, read_mask_only) if (has_mask or read_mask_only) \
                    else _load_image(img_path, white_background)
            infos.append(CameraInfo(idx, R, T, FovY, FovX, img_path, width, height, view, step, image))
            idx += 1
    return {"cam_infos": infos, "views": views, "steps": steps_used}      # (dataset_readers.py:329)


class DiskCamera(object):
    """cameras.py:17-57 (Camera / PhysCamera): world_view_transform, full_proj_transform, camera_center in the row-vector
    convention the rasterizer expects - anything get_rasterizer() accepts."""

    def __init__(self, info: CameraInfo, device="cpu", znear: float = 0.01, zfar: float = 100.0, trans=(0.0, 0.0, 0.0),
                 scale: float = 1.0):
        self.uid, self.view, self.step = info.uid, info.view, info.step
        self.R, self.T, self.FoVx, self.FoVy = info.R, info.T, info.FovX, info.FovY
        self.image_width, self.image_height = int(info.width), int(info.height)
        self.original_image = None if info.image is None else torch.tensor(info.image, device=device).permute(2, 0, 1).contiguous()
        Rt = np.zeros((4, 4))                                   # graphics_utils.py:38-49 getWorld2View2
        Rt[:3, :3] = info.R.transpose()
        Rt[:3, 3] = info.T
        Rt[3, 3] = 1.0
        C2W = np.linalg.inv(Rt)
        C2W[:3, 3] = (C2W[:3, 3] + np.asarray(trans)) * scale
        Rt = np.linalg.inv(C2W)
        self.world_view_transform = torch.tensor(Rt, dtype=torch.float32).transpose(0, 1).to(device)
        tanx, tany = math.tan(self.FoVx / 2), math.tan(self.FoVy / 2)          # graphics_utils.py:51-71 getProjectionMatrix
        top, right = tany * znear, tanx * znear
        P = torch.zeros(4, 4)
        P[0, 0] = 2.0 * znear / (2 * right)
        P[1, 1] = 2.0 * znear / (2 * top)
        P[3, 2] = 1.0
        P[2, 2] = zfar / (zfar - znear)
        P[2, 3] = -(zfar * znear) / (zfar - znear)
        self.projection_matrix = P.transpose(0, 1).to(device)
        self.full_proj_transform = (self.world_view_transform.unsqueeze(0).bmm(self.projection_matrix.unsqueeze(0))).squeeze(0)
        self.camera_center = self.world_view_transform.inverse()[3, :3]
