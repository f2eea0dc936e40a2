"""Asset preparation that sits in front of both drivers: counterpart of `prepare_simulation_data`
(/root/reference/modules/tune/utils.py:211-320).  From a reconstructed 3DGS `point_cloud.ply` and a particle point cloud
it writes, into experiments/assets/<sim_data_name>/:
    kernels.ply     Gaussians with opacity > opacity_thres (property order of gaussian_model.py:189-220)
    particles.ply   the simulated particles: the given ones (randomly permuted, every `downsample_factor`-th kept) plus the
                    centres of the Gaussians no particle fell into
    bindings.pt     sparse (K x N) binding weights + per-Gaussian particle counts
The O(K N) Mahalanobis loop of binding_utils.py runs as the grid-hashed HIP search of neuma_amd.binding.
Particles out of a mesh (particle_data.mesh_path, tune/utils.py:49-200): 'volumetric' / 'uniform' are sampled here with a
ray-parity inside test (io.sample_mesh_points; the reference calls trimesh and a prebuilt `VolumeSampling` ELF); a
particles.ply already lying in the asset folder is used as it is."""
from pathlib import Path
from typing import Optional

import torch

from . import io as nio
from .extras import mesh_sampling as mesh      # (outside the section-8 scope: see its header)
from .binding import prepare_bindings


@torch.no_grad()
def prepare_simulation_data(save_dir: Path, kernels_path: Path, particles_path: Optional[Path] = None, mesh_path: Optional[Path] = None,
                            mesh_sample_mode: str = "volumetric", mesh_sample_resolution: int = 30, sh_degree: int = 3,
                            opacity_thres: float = 0.02, particles_downsample_factor: int = 3, confidence: float = 0.95,
                            max_particles: int = 10, device="cuda") -> None:
    save_dir = Path(save_dir)
    done = all((save_dir / n).is_file() for n in ("kernels.ply", "particles.ply", "bindings.pt"))
    print("===================================")
    if done:
        print("Data already prepared. Skipping data preparation.\n")
        print("===================================\n")
        return
    print("Start preparing data for simulation.\n")
    save_dir.mkdir(parents=True, exist_ok=True)
    gaussians = nio.load_gaussians_ply(kernels_path, sh_degree, device=device)
    retain = (gaussians.get_opacity.squeeze(-1) > opacity_thres).cpu().numpy()
    print(f"Gaussians after pruning low opacity kernels: {int(retain.sum())}")
    gaussians = nio.load_gaussians_ply(kernels_path, sh_degree, device=device, mask=retain)
    nio.save_gaussians_ply(gaussians, save_dir / "kernels.ply")
    if particles_path is not None:
        print(f"Extracting particles from pcd file [{particles_path}] ...")
        particles = nio.load_particles_ply(particles_path)
    elif mesh_path is not None and (save_dir / "particles.ply").is_file():
        # particles sampled elsewhere (e.g. with the reference's tools) and placed in the asset folder: used as they are
        print(f"Using the particles found in [{save_dir / 'particles.ply'}] ...")
        particles = nio.load_particles_ply(save_dir / "particles.ply")
        particles_downsample_factor = 1
    elif mesh_path is not None:
        print(f"Sampling particles inside mesh [{mesh_path}] ({mesh_sample_mode}, resolution {mesh_sample_resolution}) ...")
        reader = mesh.read_obj_mesh if Path(mesh_path).suffix.lower() == ".obj" else mesh.read_ply_mesh
        particles = mesh.sample_mesh_points(*reader(mesh_path), mode=mesh_sample_mode, resolution=int(mesh_sample_resolution))
        particles_downsample_factor = 1
    else:
        raise ValueError("Either 'particles_path' or 'mesh_path' must be provided.")
    particles = torch.as_tensor(particles, dtype=torch.float32, device=device)
    particles, B, n_particles = prepare_bindings(gaussians, particles, confidence=confidence, max_particles=max_particles,
                                                 particles_downsample_factor=int(particles_downsample_factor), save_dir=save_dir)
    print(f"COO:  {tuple(B.indices().shape)}")
    print("\nData preparation done.")
    print("===================================\n")
