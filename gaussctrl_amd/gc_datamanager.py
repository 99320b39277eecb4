"""Data manager: the (camera, batch) source of the splat optimisation and the owner of `train_data`, the list of per-view dicts
through which the two halves of the hot path meet (image, unedited_image, depth_image, z_0_image, mask_image, image_idx --
/root/reference/gaussctrl/gc_pipeline.py:268-274,234).

Host I/O is out of scope; what the pipeline relies on is kept:
  * config fields of gc_datamanager.py:54-66 (patch_size, subset_num, sampled_views_every_subset, load_all);
  * the 4 x 10 view sub-sampling of gc_datamanager.py:90-110 (anchors every view_num // subset_num, `random.sample` per subset);
  * `next_train(step)`: a random not-yet-seen view, refilled when exhausted, batch dict copied, camera metadata cam_idx
    (gc_datamanager.py:213-235).
Under nerfstudio the class derives from FullImageDatamanager (image caching / undistortion stay nerfstudio's); stand-alone it is
fed cameras (+ images) directly."""
from __future__ import annotations

import random
from dataclasses import dataclass, field

from .ns_compat import HAVE_NERFSTUDIO, Cameras


def sample_views(view_num: int, subset_num: int, per_subset: int, rng=random) -> list:
    """gc_datamanager.py:95-104: split [0, view_num) at multiples of view_num // subset_num (first 4 anchors), draw `per_subset`
    sorted indices from each part with random.sample (driven by nerfstudio's global seed)."""
    anchors = list(range(0, view_num, view_num // subset_num))[:4] + [view_num]
    out = []
    for a, b in zip(anchors[:-1], anchors[1:]):
        out += sorted(rng.sample(list(range(a, b)), per_subset))
    return out


class _NextTrainMixin:
    def _init_sampling(self, n: int):
        self.train_unseen_cameras = list(range(n))

    view_sync = None        # optional callable(local draw or None) -> view index every rank uses (GaussCtrlPipeline, train_mode "parity")

    def _pop_view(self) -> int:
        if self.view_sync is not None:
            return self.view_sync(self._draw_view)
        return self._draw_view()

    def _draw_view(self) -> int:
        i = self.train_unseen_cameras.pop(random.randint(0, len(self.train_unseen_cameras) - 1))
        if len(self.train_unseen_cameras) == 0:
            self.train_unseen_cameras = list(range(len(self.train_data)))
        return i


if HAVE_NERFSTUDIO:  # pragma: no cover - executed with nerfstudio (or tests/fake_nerfstudio) on the path
    from nerfstudio.data.datamanagers.full_images_datamanager import FullImageDatamanager, FullImageDatamanagerConfig  # type: ignore

    @dataclass
    class GaussCtrlDataManagerConfig(FullImageDatamanagerConfig):
        _target: type = field(default_factory=lambda: GaussCtrlDataManager)
        patch_size: int = 32
        subset_num: int = 4
        sampled_views_every_subset: int = 10
        load_all: bool = False

    class GaussCtrlDataManager(_NextTrainMixin, FullImageDatamanager):
        config: GaussCtrlDataManagerConfig

        def __init__(self, config, device="cpu", test_mode="val", world_size=1, local_rank=0, **kwargs):
            super().__init__(config, device, test_mode, world_size, local_rank, **kwargs)
            n_all = len(self.train_dataset)
            want = config.subset_num * config.sampled_views_every_subset
            if n_all <= want or config.load_all:
                self.cameras = self.train_dataset.cameras
                self.train_data = self.cached_train
                self._subsampled = False
            else:
                idx = sample_views(n_all, config.subset_num, config.sampled_views_every_subset)
                self.cameras = [self.train_dataset.cameras[i:i + 1] for i in idx]
                self.train_data = []
                for j, i in enumerate(idx):
                    d = self.cached_train[i]
                    d["image_idx"] = j
                    self.train_data.append(d)
                self._subsampled = True
            self._init_sampling(len(self.train_data))

        def next_train(self, step: int):
            i = self._pop_view()
            data = dict(self.train_data[i])
            data["image"] = data["image"].to(self.device)
            camera = (self.cameras[i] if self._subsampled else self.cameras[i:i + 1]).to(self.device)
            if camera.metadata is None:
                camera.metadata = {}
            camera.metadata["cam_idx"] = i
            return camera, data
else:
    @dataclass
    class GaussCtrlDataManagerConfig:
        """gc_datamanager.py:54-66"""
        _target: type = field(default_factory=lambda: GaussCtrlDataManager)
        patch_size: int = 32
        subset_num: int = 4
        sampled_views_every_subset: int = 10
        load_all: bool = False

        def setup(self, cameras=None, images=None, **kw):
            return self._target(self, cameras=cameras, images=images)

    class GaussCtrlDataManager(_NextTrainMixin):
        """Stand-alone: holds `cameras` (a Cameras batch; indexable per view) and `train_data`."""

        def __init__(self, config=None, cameras: Cameras = None, images=None, seed: int | None = None, **kw):
            self.config = config or GaussCtrlDataManagerConfig()
            n_all = len(cameras)
            want = self.config.subset_num * self.config.sampled_views_every_subset
            idx = list(range(n_all))
            if n_all > want and not self.config.load_all:
                idx = sample_views(n_all, self.config.subset_num, self.config.sampled_views_every_subset)
            self.view_indices = idx
            self.cameras = [cameras[i] for i in idx] if idx != list(range(n_all)) else cameras
            self.train_data = [{"image_idx": j, "image": None if images is None else images[i]} for j, i in enumerate(idx)]
            if seed is not None:
                random.seed(seed)
            self._init_sampling(len(self.train_data))

        def next_train(self, step: int):
            i = self._pop_view()
            cam = self.cameras[i]
            if getattr(cam, "metadata", None) is None:
                cam.metadata = {}
            cam.metadata["cam_idx"] = i
            return cam, dict(self.train_data[i])


class SimpleDataManager(GaussCtrlDataManager if not HAVE_NERFSTUDIO else _NextTrainMixin):
    """cameras (+ images) in, no I/O: `SimpleDataManager(cameras, images=None, seed=0)` (tests, bench, stand-alone use)."""

    def __init__(self, cameras: Cameras, images=None, seed: int = 0, load_all: bool = True):
        if HAVE_NERFSTUDIO:      # the nerfstudio-backed class needs a dataparser: keep the plain holder
            self.cameras = cameras
            self.train_data = [{"image_idx": i, "image": None if images is None else images[i]} for i in range(len(cameras))]
            self._init_sampling(len(self.train_data))
            return
        cfg = GaussCtrlDataManagerConfig(load_all=load_all)
        super().__init__(cfg, cameras=cameras, images=images, seed=seed)

    if HAVE_NERFSTUDIO:      # plain holder: what GaussCtrlPipeline / VanillaPipeline ask of a data manager
        def next_train(self, step: int):
            i = self._pop_view()
            cam = self.cameras[i]
            if getattr(cam, "metadata", None) is None:
                cam.metadata = {}
            cam.metadata["cam_idx"] = i
            return cam, dict(self.train_data[i])

        def get_training_callbacks(self, training_callback_attributes):
            return []

        def get_param_groups(self):
            return {}
