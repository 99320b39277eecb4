"""CPU tests of the plugin surface:
  * every default of the trainer / pipeline / datamanager / model configs and every optimizer group is diff-checked against
    tests/golden/ref_config.json, which tests/golden/make_config_golden.py extracts from the reference's own sources
    (/root/reference/gaussctrl/gc_config.py:40-92, gc_pipeline.py:48-73, gc_datamanager.py:54-66, gc_model.py:39-50,
    gc_trainer.py:42-47, pyproject.toml:38-42);
  * with a nerfstudio package importable (tests/fake_nerfstudio: a test double of the few bases the plugin subclasses) the entry
    point yields a MethodSpecification whose .config is a real TrainerConfig subclass instance holding config OBJECTS, the model /
    pipeline / datamanager / trainer classes derive from nerfstudio's, and the trainer drives render_reverse -> edit_images ->
    render_rate iterations;
  * the stand-alone optimizers follow the exponential-decay schedules."""
import json
import math
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_config.json")))


def _fields(obj):
    import dataclasses
    return {f.name: getattr(obj, f.name) for f in dataclasses.fields(obj)}


def test_trainer_flags_match_reference():
    from gaussctrl_amd import gc_config
    ref = REF["method_specification"]["config"]
    cfg = gc_config.gaussctrl_method.config
    for k, v in ref.items():
        if k in ("__call__", "pipeline", "optimizers", "viewer"):
            continue
        assert getattr(cfg, k) == v, (k, getattr(cfg, k), v)
    assert gc_config.gaussctrl_method.description == REF["method_specification"]["description"]
    assert gc_config.VIEWER == {k: v for k, v in ref["viewer"].items() if k != "__call__"}
    # the class default of GaussCtrlTrainerConfig itself (gc_trainer.py:42-47)
    from gaussctrl_amd.gc_trainer import GaussCtrlTrainerConfig
    for k, v in REF["GaussCtrlTrainerConfig"]["fields"].items():
        assert GaussCtrlTrainerConfig.__dataclass_fields__[k].default == v


def test_optimizer_groups_match_reference():
    from gaussctrl_amd import gc_config
    ref = REF["method_specification"]["config"]["optimizers"]
    ours = gc_config.optimizer_table()
    assert set(ours) == set(ref)
    for name, r in ref.items():
        assert ours[name]["optimizer"]["lr"] == pytest.approx(r["optimizer"]["lr"], rel=1e-12)
        assert ours[name]["optimizer"]["eps"] == r["optimizer"]["eps"]
        if r["scheduler"] is None:
            assert ours[name]["scheduler"] is None
        else:
            assert ours[name]["scheduler"] == {k: v for k, v in r["scheduler"].items() if k != "__call__"}


@pytest.mark.parametrize("cls_name,mod", [("GaussCtrlPipelineConfig", "gc_pipeline"), ("GaussCtrlDataManagerConfig", "gc_datamanager"),
                                          ("GaussCtrlModelConfig", "gc_model")])
def test_config_defaults_match_reference(cls_name, mod):
    import importlib
    cls = getattr(importlib.import_module("gaussctrl_amd." + mod), cls_name)
    ours = _fields(cls())
    for k, v in REF[cls_name]["fields"].items():
        assert k in ours, f"{cls_name}.{k} missing"
        if isinstance(v, dict) and "__expr__" in v:
            assert type(ours[k]).__name__ == v["__expr__"].split("(")[0]
        else:
            assert ours[k] == v and type(ours[k]) is type(v), (cls_name, k, ours[k], v)


def test_entry_points_match_reference():
    import re
    pp = open(os.path.join(ROOT, "pyproject.toml")).read()
    eps = dict(re.findall(r"^\s*([\w-]+)\s*=\s*['\"]([\w.:]+)['\"]\s*$", pp.split("[project.entry-points", 1)[1], flags=re.M))
    for name, target in REF["entry_points"].items():
        assert eps[name] == target.replace("gaussctrl.", "gaussctrl_amd.", 1)
    from gaussctrl_amd import gc_config, gc_render
    assert hasattr(gc_config, "gaussctrl_method") and callable(gc_render.entrypoint)


def test_exponential_decay_schedule():
    from gaussctrl_amd.gc_config import scheduled_lr
    assert scheduled_lr("xyz", 0) == pytest.approx(1.6e-4)
    assert scheduled_lr("xyz", 15000) == pytest.approx(math.sqrt(1.6e-4 * 1.6e-6))
    assert scheduled_lr("xyz", 30000) == pytest.approx(1.6e-6) == scheduled_lr("xyz", 30499)      # a GaussCtrl run: steps 30000..30499
    assert scheduled_lr("camera_opt", 30000) == pytest.approx(5e-5)
    assert scheduled_lr("opacity", 30000) == 0.05


_NS_SCRIPT = r'''
import dataclasses, json, sys
import nerfstudio                                  # the test double
from nerfstudio.engine.trainer import Trainer, TrainerConfig
from nerfstudio.pipelines.base_pipeline import VanillaPipeline, VanillaPipelineConfig
from nerfstudio.models.splatfacto import SplatfactoModel, SplatfactoModelConfig
from nerfstudio.data.datamanagers.full_images_datamanager import FullImageDatamanager, FullImageDatamanagerConfig
from nerfstudio.engine.optimizers import AdamOptimizerConfig
from nerfstudio.engine.schedulers import ExponentialDecaySchedulerConfig
from nerfstudio.plugins.types import MethodSpecification
from gaussctrl_amd import gc_config, ns_compat
from gaussctrl_amd.gc_model import GaussCtrlModel, GaussCtrlModelConfig
from gaussctrl_amd.gc_pipeline import GaussCtrlPipeline, GaussCtrlPipelineConfig
from gaussctrl_amd.gc_datamanager import GaussCtrlDataManager, GaussCtrlDataManagerConfig
from gaussctrl_amd.gc_trainer import GaussCtrlTrainer, GaussCtrlTrainerConfig
assert ns_compat.HAVE_NERFSTUDIO
spec = gc_config.gaussctrl_method
assert isinstance(spec, MethodSpecification) and isinstance(spec.config, TrainerConfig) and isinstance(spec.config, GaussCtrlTrainerConfig)
assert spec.config.method_name == "gaussctrl"          # what nerfstudio's plugin discovery reads
assert issubclass(GaussCtrlTrainer, Trainer) and issubclass(GaussCtrlPipeline, VanillaPipeline)
assert issubclass(GaussCtrlModel, SplatfactoModel) and issubclass(GaussCtrlDataManager, FullImageDatamanager)
assert isinstance(spec.config.pipeline, VanillaPipelineConfig) and isinstance(spec.config.pipeline.model, SplatfactoModelConfig)
assert isinstance(spec.config.pipeline.datamanager, FullImageDatamanagerConfig) and spec.config.pipeline.datamanager.dataparser.load_3D_points is True
assert spec.config.pipeline.datamanager._target is GaussCtrlDataManager and spec.config.pipeline._target is GaussCtrlPipeline
o = spec.config.optimizers
assert isinstance(o["xyz"]["optimizer"], AdamOptimizerConfig) and isinstance(o["xyz"]["scheduler"], ExponentialDecaySchedulerConfig)
assert o["features_dc"]["scheduler"] is None and o["camera_opt"]["scheduler"].lr_final == 5e-5
# drive the trainer: the pipeline class is swapped for a recorder that skips the GPU networks but keeps the real __init__ contract
calls = []
class Rec(GaussCtrlPipeline):
    def __init__(self, config, device, test_mode="val", world_size=1, local_rank=0, grad_scaler=None):
        VanillaPipeline.__init__(self, config, device, test_mode, world_size, local_rank)
        self.test_mode = test_mode
    def render_reverse(self): calls.append("render_reverse")
    def edit_images(self): calls.append("edit_images")
cfg = dataclasses.replace(spec.config)
cfg.pipeline = dataclasses.replace(cfg.pipeline, _target=Rec, render_rate=7)
tr = cfg.setup(local_rank=0, world_size=1)
assert isinstance(tr, GaussCtrlTrainer)
tr.setup(test_mode="val")
assert calls == ["render_reverse", "edit_images"], calls
assert isinstance(tr.pipeline.model, GaussCtrlModel) and isinstance(tr.pipeline.datamanager, GaussCtrlDataManager)
assert len(tr.pipeline.datamanager.train_data) == 40 and len(tr.pipeline.datamanager.cameras) == 40      # 4 x 10 of the 57 fake views
cam, batch = tr.pipeline.datamanager.next_train(0)
assert cam.metadata["cam_idx"] in range(40) and batch["image"].shape == (32, 32, 3)
assert len(tr.callbacks) >= 1                           # SplatfactoModel's callbacks reach the trainer through the pipeline
tr.train()
assert tr.trained_steps == list(range(30000, 30007)) and tr.config.max_num_iterations == 1000
print("NS-OK")
'''


def test_nerfstudio_branches_with_test_double():
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "tests", "fake_nerfstudio"), ROOT, env.get("PYTHONPATH", "")])
    r = subprocess.run([sys.executable, "-c", _NS_SCRIPT], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "NS-OK" in r.stdout, r.stdout[-3000:]


def test_standalone_trainer_sequence_cpu():
    """Stand-alone trainer: setup() calls render_reverse then edit_images (test_mode 'val'), train() runs render_rate iterations from
    step 30000 with the scheduled learning rates and the gradient-accumulation rule of gc_trainer.py:265-281."""
    import torch
    from gaussctrl_amd import gc_config
    from gaussctrl_amd.gc_trainer import GaussCtrlTrainer
    calls = []

    class FakeModel(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.ones(4))
            self.step = 0

        def get_param_groups(self):
            return {"xyz": [self.p]}

    class FakePipe:
        test_mode = "val"

        def __init__(self):
            self.model = FakeModel()
            self.config = type("C", (), {"render_rate": 5})()

        def render_reverse(self): calls.append("rr")
        def edit_images(self): calls.append("ei")
        def get_training_callbacks(self, attrs): return []
        def train(self): pass
        def reduce_gradients(self): pass

        def get_train_loss_dict(self, step):
            calls.append(step)
            return None, {"main_loss": (self.model.p ** 2).sum()}, {}

    pipe = FakePipe()
    cfg = gc_config.gaussctrl_method.config
    cfg = type(cfg)(**{**{f: getattr(cfg, f) for f in cfg.__dataclass_fields__}, "pipeline": type("PC", (), {"setup": lambda self, **kw: pipe})()})
    tr = GaussCtrlTrainer(cfg, device="cpu")
    tr.setup_optimizers = lambda: {"xyz": torch.optim.Adam([pipe.model.p], lr=1.0)}
    tr.setup("val")
    assert calls == ["rr", "ei"]
    tr.train()
    assert calls[2:] == list(range(30000, 30005))
    assert tr.optimizers["xyz"].param_groups[0]["lr"] == pytest.approx(1.6e-6)      # exp-decay value at step >= 30000
