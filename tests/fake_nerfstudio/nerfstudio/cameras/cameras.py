import torch


class Cameras:
    def __init__(self, camera_to_worlds, fx, fy, cx, cy, width, height, metadata=None):
        c2w = torch.as_tensor(camera_to_worlds, dtype=torch.float32)
        if c2w.dim() == 2:
            c2w = c2w[None]
        n = c2w.shape[0]
        col = lambda v, dt: torch.as_tensor(v, dtype=dt).reshape(-1, 1).expand(n, 1).clone()
        self.camera_to_worlds = c2w[:, :3, :4].contiguous()
        self.fx, self.fy, self.cx, self.cy = (col(v, torch.float32) for v in (fx, fy, cx, cy))
        self.width, self.height = col(width, torch.int64), col(height, torch.int64)
        self.metadata = metadata

    @property
    def shape(self):
        return self.camera_to_worlds.shape[:1]

    def __len__(self):
        return self.camera_to_worlds.shape[0]

    def __getitem__(self, i):
        if isinstance(i, int):
            i = slice(i, i + 1)
        return Cameras(self.camera_to_worlds[i], self.fx[i], self.fy[i], self.cx[i], self.cy[i], self.width[i], self.height[i], self.metadata)

    def to(self, device):
        return self
