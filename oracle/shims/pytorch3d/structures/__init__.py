"""pytorch3d.structures.Volumes — only what models/rotate.py:50-51 and models/volume_render.py:59
use: densities/features storage, get_coord_grid(world_coordinates=True), world_to_local_coords.
Local coords span [-1,1] between the first and last VOXEL CENTRES (align_corners=True
convention); local->world = scale by 0.5*(res-1)*voxel_size (volume_translation = 0)."""
import torch


class Volumes:
    def __init__(self, densities, features=None, voxel_size=1.0, volume_translation=(0.0, 0.0, 0.0)):
        self._densities = densities
        self._features = features
        n, _, d, h, w = densities.shape
        self._res_xyz = torch.tensor([w, h, d], dtype=torch.float32, device=densities.device)
        vs = torch.as_tensor(voxel_size, dtype=torch.float32, device=densities.device)
        if vs.dim() == 0:
            vs = vs.expand(3)
        self._voxel_size = vs
        self._n = n
        self._dhw = (d, h, w)

    def densities(self):
        return self._densities

    def features(self):
        return self._features

    def _half_extent(self):
        return 0.5 * (self._res_xyz - 1.0) * self._voxel_size   # [3] (x,y,z)

    def get_coord_grid(self, world_coordinates=True):
        d, h, w = self._dhw
        dev = self._densities.device
        zs = torch.linspace(-1.0, 1.0, d, device=dev)
        ys = torch.linspace(-1.0, 1.0, h, device=dev)
        xs = torch.linspace(-1.0, 1.0, w, device=dev)
        Z, Y, X = torch.meshgrid(zs, ys, xs, indexing="ij")
        grid = torch.stack([X, Y, Z], dim=-1)[None].expand(self._n, d, h, w, 3)
        if world_coordinates:
            grid = grid * self._half_extent().to(dev)
        return grid

    def world_to_local_coords(self, points_3d_world):
        return points_3d_world / self._half_extent().to(points_3d_world.device)
